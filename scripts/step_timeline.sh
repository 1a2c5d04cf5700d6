#!/bin/bash
# usage on the GPU box: bash scripts/step_timeline.sh [landmarks]   (environment knobs are inherited)
N=${1:-256}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/stl
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/stl -o t -- python3 $ROOT/bench.py --steps 96 --warmup 10 --landmarks $N --no-cpu-baseline --no-full-loop > $ROOT/gpurun_out/stl.log 2>&1 || { tail -5 $ROOT/gpurun_out/stl.log; exit 1; }
python3 $ROOT/scripts/step_timeline.py /tmp/stl
