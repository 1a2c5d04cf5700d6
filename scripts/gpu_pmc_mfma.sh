#!/bin/bash
# MFMA-utilisation counters of the hot kernels: one rocprofv3 --pmc pass (SQ block, <= 8 counters) over a short bench
# run, kernel-trace only (no other trace domains), the program itself behind "--".
# Usage on the GPU box: bash scripts/gpu_pmc_mfma.sh <tag>
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_mfma -o p -- python3 $ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-full-loop > $OUT/pmc_${TAG}_mfma.log 2>&1
rc=$?
echo "pmc mfma rc=$rc"
if [ $rc -ne 0 ]; then tail -8 $OUT/pmc_${TAG}_mfma.log; fi
find $OUT/pmc_${TAG}_mfma -name "*counter_collection*" | head
exit $rc
