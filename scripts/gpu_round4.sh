#!/bin/bash
# Round-4 measurement visit: bench (default line), rocprofv3 kernel stats, the three PMC passes, N=1024 bench + stats, stamps.
# Usage on the GPU box: bash scripts/gpu_round4.sh r04
set -o pipefail
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; tail -2 $OUT/bench_$TAG.err
python scripts/persist_stamps.py 256 > $OUT/stamps_$TAG.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o ekfvio -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-full-loop > $OUT/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
cd $ROOT
bash scripts/gpu_pmc.sh $TAG && bash scripts/gpu_pmc_mfma.sh $TAG
bash scripts/step_timeline.sh 256 > $OUT/timeline_$TAG.txt 2>&1; tail -6 $OUT/timeline_$TAG.txt
timeout -k 10 300 python bench.py --landmarks 1024 --steps 40 --warmup 6 --no-full-loop > $OUT/bench_${TAG}_n1024.json 2> $OUT/bench_${TAG}_n1024.err; echo "bench1024 rc=$?"
bash scripts/n1024_prof.sh $TAG 2>&1 | tail -12
