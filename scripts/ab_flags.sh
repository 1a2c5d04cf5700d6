#!/bin/bash
# Same-box A/B of a compile-time switch: bash scripts/ab_flags.sh "<flags A>" "<flags B>" [bench args]
# Builds the library with each flag set in turn (A, B, A, B) and runs bench.py each time; prints steps/s.
A="$1"; B="$2"; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for v in A B A B; do
  if [ $v = A ]; then FL="$A"; else FL="$B"; fi
  EKFVIO_EXTRA_HIPCC_FLAGS="$FL" python -c "import ekf_vio_amd._build as b; b.build(force=True)" > gpurun_out/ab_build.log 2>&1 || { tail -5 gpurun_out/ab_build.log; exit 1; }
  # (the same flags in the run's environment: the library's staleness check compares them with the build's)
  EKFVIO_EXTRA_HIPCC_FLAGS="$FL" timeout -k 10 200 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-full-loop "$@" > gpurun_out/ab_run.json 2> gpurun_out/ab_run.err || exit 1
  python -c "
import json
j=json.load(open('gpurun_out/ab_run.json')); print('$v [$FL]', round(j['value'],1), {k: round(x,1) for k,x in j['stage_us_per_step'].items()})"
done
