#!/bin/bash
# Per-kernel durations of several builds on the same box: rocprofv3 kernel statistics of scripts/ab_lib.py, one run per library.
# usage: bash scripts/ab_prof.sh out_prefix lib1.so lib2.so ...
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$1; shift
for L in "$@"; do
  B=$(basename $L .so)
  rm -rf /tmp/abp_$B
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/abp_$B -o x --output-format csv -- python3 $R/scripts/ab_lib.py $R/$L > $R/gpurun_out/${P}_$B.log 2>&1
  F=$(find /tmp/abp_$B -name "*kernel_stats.csv" | head -1)
  echo "== $B" >> $R/gpurun_out/${P}_summary.txt
  python3 - "$F" >> $R/gpurun_out/${P}_summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("%-48s calls %6s avg %9.1f ns" % (r["Name"].replace("(anonymous namespace)::", "")[:48], r["Calls"], float(r["AverageNs"])))
PY
done
cat $R/gpurun_out/${P}_summary.txt
