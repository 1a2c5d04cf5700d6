#!/bin/bash
# usage on the GPU box: bash scripts/frame_timeline.sh [landmarks] [outputs 0/1]
N=${1:-64}; OUTS=${2:-1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ftl
cat > /tmp/ftl_run.py <<PY
import sys
sys.path.insert(0, "$ROOT")
import bench
r = bench.full_loop($N, 0, frames=30, warm=6, outputs=bool($OUTS))
print(r["frames_per_s"], r["ms_per_frame"])
PY
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ftl -o t -- python3 /tmp/ftl_run.py > $ROOT/gpurun_out/ftl.log 2>&1 || { tail -5 $ROOT/gpurun_out/ftl.log; exit 1; }
python3 $ROOT/scripts/frame_timeline.py /tmp/ftl
