#!/bin/bash
# Tile order of the mirrored second Joseph GEMM (gemm.hip, GemmEpi::sym_w): steps/s, the kernel's mean duration and its FETCH_SIZE per strip width.
# (needs the EKFVIO_SYM_W experiment switch of its commit in gemm.hip launch_gemm_cfg; kept as the record of how the strip width was chosen)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in $1; do
  export EKFVIO_SYM_W=$w
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/symw_$w -o p -- python3 $ROOT/bench.py --landmarks 1024 --steps 12 --warmup 3 --no-cpu-baseline --no-full-loop --replay-only > /dev/null 2>&1
  python3 - <<PY
import csv, collections
f, d = collections.defaultdict(list), collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/symw_$w/p_counter_collection.csv")):
    if r["Counter_Name"] == "FETCH_SIZE" and "gemm_f32_mfma_kernel<true, 1, " in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0][-12:]
        f[k].append(float(r["Counter_Value"]))
        d[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("sym_w $w", {k: ("%.0f MB" % (2 * 1024 * sum(v) / len(v) / 1e6), "%.1f us" % (sum(d[k]) / len(d[k]) / 1e3)) for k, v in f.items()})
PY
done
cd $ROOT
for rep in 1 2; do for w in $1; do EKFVIO_SYM_W=$w python bench.py --steps 64 --warmup 10 --landmarks 1024 --no-cpu-baseline --no-full-loop 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('sym_w $w steps/s %.1f' % r['value'], 'gemm_update %.1f' % r['stage_us_per_step']['gemm_update'])"; done; done
