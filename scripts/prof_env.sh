#!/bin/bash
# rocprofv3 kernel statistics of the bench for two settings of one environment knob, same box:
#   bash scripts/prof_env.sh NAME A B [landmarks] -> gpurun_out/prof_NAME_A.csv, gpurun_out/prof_NAME_B.csv (kernel, calls, mean us)
NAME=$1; A=$2; B=$3; N=${4:-256}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $A $B; do
  export $NAME=$v
  rm -rf $OUT/prof_${NAME}_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${NAME}_$v -o p -- python3 $ROOT/bench.py --steps 100 --warmup 10 --landmarks $N --no-cpu-baseline --no-full-loop > $OUT/prof_${NAME}_$v.log 2>&1 || { tail -5 $OUT/prof_${NAME}_$v.log; exit 1; }
  f=$(find $OUT/prof_${NAME}_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$NAME=$v" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(sys.argv[2])
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("  %-58s calls %6s  mean %8.2f us  total %6.1f %%" % (r["Name"].replace("(anonymous namespace)::", "")[:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
