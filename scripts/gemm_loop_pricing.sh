#!/bin/bash
# Prices the parts of gemm16_kernel's K loop: rebuilds the library with pieces compiled out (-DG16X_NO*) and runs the K sweep.
# Results are wrong in those builds; run on the GPU box only (it rebuilds ekf_vio_amd/lib in place: rebuild normally afterwards).
for fl in "" "-DG16X_NOLOAD=1" "-DG16X_NOLOAD=1 -DG16X_NOSTAGE=1" "-DG16X_NOLOAD=1 -DG16X_NOSTAGE=1 -DG16X_NOBAR=1" "-DG16X_NOFR=1" "-DG16X_NOLOAD=1 -DG16X_NOSTAGE=1 -DG16X_NOBAR=1 -DG16X_NOFR=1"; do
  EKFVIO_EXTRA_HIPCC_FLAGS="$fl" python -c "from ekf_vio_amd import _build; _build.build(force=True)" > /dev/null 2>&1
  echo "flags: $fl"
  timeout -k 10 120 python scripts/gemm_sweep.py 48 | head -3 | tail -2
done
python -c "from ekf_vio_amd import _build; _build.build(force=True)" > /dev/null 2>&1  # back to the production build
