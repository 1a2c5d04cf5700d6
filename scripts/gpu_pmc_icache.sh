#!/bin/bash
# Instruction-cache counters of the step's kernels (are the once-through, fully unrolled chains fetching from L2?):
# one rocprofv3 --pmc pass over scripts/ab_lib.py.  Usage on the GPU box: bash scripts/gpu_pmc_icache.sh <tag> <lib.so>
set -o pipefail
TAG=${1:-ic}
LIB=${2:-ekf_vio_amd/lib/libekfvio_hip.so}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/pmc_avail.txt 2>&1
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQC_TC_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" $OUT/pmc_avail.txt | sort -u | tr '\n' ' '
echo
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH --kernel-trace --output-format csv -d $OUT/pmc_${TAG} -o p -- python3 $ROOT/scripts/ab_lib.py $ROOT/$LIB > $OUT/pmc_${TAG}.log 2>&1
rc=$?
echo "pmc rc=$rc"; tail -3 $OUT/pmc_${TAG}.log
python3 - $OUT/pmc_${TAG} <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); sys.exit(0)
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].strip()
    per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in per.items():
    print("%-40s n=%5d " % (k[:40], len(next(iter(c.values())))) + "  ".join("%s %.0f" % (n, sum(v) / len(v)) for n, v in sorted(c.items())))
PY
exit 0
