#!/bin/bash
# Round-6 measurement visit: GPU tests, bench (default line, carries config 3), rocprofv3 kernel stats at N = 256 and N = 1024 (replayed steps only), the PMC
# traffic passes at N = 256, the MFMA-busy pass, the persistent launch's stamps with and without T2 inside.  Usage on the GPU box: bash scripts/gpu_round6.sh r06 [notests]
set -o pipefail
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
if [ "${2:-}" != "notests" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/${TAG}_pytest_gpu.log
fi
timeout -k 10 500 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; tail -2 $OUT/bench_$TAG.err
for v in 0 1; do EKFVIO_T2=$v python scripts/t2_stamps.py 256 > $OUT/t2_stamps_${TAG}_$v.txt 2>&1; done
EKFVIO_T2=1 python scripts/persist_stamps.py 256 > $OUT/stamps_$TAG.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o ekfvio -- python3 $ROOT/bench.py --steps 100 --warmup 10 --landmarks 256 --no-cpu-baseline --no-full-loop --replay-only > $OUT/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_n1024 -o ekfvio -- python3 $ROOT/bench.py --steps 40 --warmup 10 --landmarks 1024 --no-cpu-baseline --no-full-loop --replay-only > $OUT/rocprof_${TAG}_n1024.log 2>&1; echo "rocprof n1024 rc=$?"
cd $ROOT
bash scripts/gpu_pmc.sh $TAG && bash scripts/gpu_pmc_mfma.sh $TAG
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_n1024_$c -o p -- python3 $ROOT/bench.py --landmarks 1024 --steps 12 --warmup 3 --no-cpu-baseline --no-full-loop > $OUT/pmc_${TAG}_n1024_$c.log 2>&1
  echo "pmc n1024 $c rc=$?"
done
cd $ROOT
