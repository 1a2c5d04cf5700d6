#!/bin/bash
# usage on the GPU box: bash scripts/node_defaults_prof.sh <tag>
set -u
TAG=${1:-nd}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ndp_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ndp_$TAG -o x --output-format csv -- python3 $R/scripts/node_defaults_prof.py > $R/gpurun_out/ndp_$TAG.log 2>&1
F=$(find /tmp/ndp_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print("%-52s calls %6s avg %9.1f ns  total %8.1f us" % (r["Name"].replace("(anonymous namespace)::", "")[:52], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"]) / 1e3))
PY
tail -1 $R/gpurun_out/ndp_$TAG.log
