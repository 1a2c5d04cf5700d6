#!/bin/bash
# Kernel statistics of the full loop (ekfvio_step_image per frame): rocprofv3 over a short bench run.
# usage on the GPU box: bash scripts/full_loop_prof.sh <tag>
set -u
TAG=${1:-fl}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/flp_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/flp_$TAG -o x --output-format csv -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/flp_$TAG.log 2>&1
F=$(find /tmp/flp_$TAG -name "*kernel_stats.csv" | head -1)
cp $F $R/gpurun_out/flp_${TAG}_kernel_stats.csv
python3 - "$F" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:24]:
    print("%-52s calls %6s avg %9.1f ns" % (r["Name"].replace("(anonymous namespace)::", "")[:52], r["Calls"], float(r["AverageNs"])))
PY
