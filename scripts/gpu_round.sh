#!/bin/bash
# One GPU-box visit: parity tests, bench, rocprofv3 kernel stats.  Usage (from repo root on the box):
#   bash scripts/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
rm -f $OUT/parity_report.jsonl
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest_gpu_$TAG.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -15 $OUT/pytest_gpu_$TAG.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
rc=$?
echo "bench rc=$rc"; cat $OUT/bench_$TAG.json; tail -3 $OUT/bench_$TAG.err
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o ekfvio -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/rocprof_$TAG.log 2>&1
rc=$?
echo "rocprof rc=$rc"; tail -3 $OUT/rocprof_$TAG.log
find $OUT/prof_$TAG -name "*stats*" | head
