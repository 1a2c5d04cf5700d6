#!/bin/bash
# Same-box A/B of compile-time switches at a given N: bash scripts/ab_flags_n.sh N "<flags A>" "<flags B>" ...
# Builds the library with each flag set in turn (twice round) and runs bench.py --landmarks N each time; prints steps/s.
N=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
  for FL in "$@"; do
    EKFVIO_EXTRA_HIPCC_FLAGS="$FL" python -c "import ekf_vio_amd._build as b; b.build(force=True)" > gpurun_out/ab_build.log 2>&1 || { tail -5 gpurun_out/ab_build.log; exit 1; }
    EKFVIO_EXTRA_HIPCC_FLAGS="$FL" timeout -k 10 200 python bench.py --steps 64 --warmup 10 --landmarks $N --no-cpu-baseline --no-full-loop > gpurun_out/ab_run.json 2> gpurun_out/ab_run.err || exit 1
    python -c "
import json
j=json.load(open('gpurun_out/ab_run.json')); print('[$FL]', round(j['value'],1), {k: round(x,1) for k,x in j['stage_us_per_step'].items()})"
  done
done
