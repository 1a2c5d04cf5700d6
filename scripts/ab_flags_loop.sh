#!/bin/bash
# Same-box A/B of compile-time switches on the image loop: bash scripts/ab_flags_loop.sh "<flags A>" "<flags B>" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
  for FL in "$@"; do
    EKFVIO_EXTRA_HIPCC_FLAGS="$FL" python -c "import ekf_vio_amd._build as b; b.build(force=True)" > gpurun_out/ab_build.log 2>&1 || { tail -5 gpurun_out/ab_build.log; exit 1; }
    echo "[$FL]"; EKFVIO_EXTRA_HIPCC_FLAGS="$FL" bash scripts/ab_loop_lib.sh 1 2>/dev/null
  done
done
