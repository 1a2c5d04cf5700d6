#!/bin/bash
# Per-launch durations of the sweep at N = 1024 (rocprofv3 kernel trace of a short bench run).  usage: bash scripts/n1024_prof.sh <tag>
set -u
TAG=${1:-n1024}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/np_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/np_$TAG -o x --output-format csv -- python3 $R/bench.py --landmarks 1024 --steps 12 --warmup 3 --no-cpu-baseline --no-full-loop --profile-steps 2 > $R/gpurun_out/np_$TAG.log 2>&1
S=$(find /tmp/np_$TAG -name "*kernel_stats.csv" | head -1); T=$(find /tmp/np_$TAG -name "*kernel_trace.csv" | head -1)
cp $S $R/gpurun_out/np_${TAG}_kernel_stats.csv
python3 - "$S" "$T" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-52s calls %6s avg %9.1f ns" % (r["Name"].replace("(anonymous namespace)::", "")[:52], r["Calls"], float(r["AverageNs"])))
rows = list(csv.DictReader(open(sys.argv[2])))
st = [r for r in rows if "chol_step" in r["Kernel_Name"]]
last = st[-31:]
print("sweep launch durations (us), last update:", " ".join("%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in last))
print("gaps to next launch (us):", " ".join("%.1f" % ((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3) for a, b in zip(last[:-1], last[1:])))
print("sweep span (us): %.1f" % ((int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3))
PY
