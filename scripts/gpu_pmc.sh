#!/bin/bash
# HBM-side traffic of the hot kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short
# bench run, kernel-trace only (no other trace domains).  Usage on the GPU box: bash scripts/gpu_pmc.sh <tag>
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$c -o p -- python3 $ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-full-loop > $OUT/pmc_${TAG}_$c.log 2>&1
  rc=$?
  echo "pmc $c rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 $OUT/pmc_${TAG}_$c.log; exit $rc; fi
done
find $OUT/pmc_${TAG}_* -name "*counter_collection*" | head
