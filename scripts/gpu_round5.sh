#!/bin/bash
# Round-5 measurement visit: GPU tests, bench (default line, carries config 3), rocprofv3 kernel stats at N = 256 and N = 1024, the PMC traffic passes at
# both sizes, the MFMA-busy pass, chain stamps.  Usage on the GPU box: bash scripts/gpu_round5.sh r05 [notests]
set -o pipefail
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
if [ "${2:-}" != "notests" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/${TAG}_pytest_gpu.log
fi
timeout -k 10 500 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; tail -2 $OUT/bench_$TAG.err
python scripts/persist_stamps.py 256 > $OUT/stamps_$TAG.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o ekfvio -- python3 $ROOT/bench.py --steps 100 --warmup 10 --landmarks 256 --no-cpu-baseline --no-full-loop > $OUT/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
cd $ROOT
bash scripts/gpu_pmc.sh $TAG && bash scripts/gpu_pmc_mfma.sh $TAG
bash scripts/n1024_prof.sh $TAG 2>&1 | tail -12
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_n1024_$c -o p -- python3 $ROOT/bench.py --landmarks 1024 --steps 12 --warmup 3 --no-cpu-baseline --no-full-loop > $OUT/pmc_${TAG}_n1024_$c.log 2>&1
  echo "pmc n1024 $c rc=$?"
done
