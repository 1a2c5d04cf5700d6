"""Summarises the two rocprofv3 --pmc passes at N = 1024 (scripts/gpu_round6.sh) into profiles/<name>_pmc_traffic_n1024.json.
Usage: python scripts/pmc_summary_n1024.py <tag> [<profile name>]   (reads gpurun_out/pmc_<tag>_n1024_{FETCH,WRITE}_SIZE/p_counter_collection.csv)"""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else tag
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
n, m_pad = 22 + 3 * 1024, 2048
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --landmarks 1024 --steps 12 "
                 "--warmup 3 --no-cpu-baseline --no-full-loop, MI355X",
       "units": "FETCH_SIZE/WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts half of the bytes of 16-B/lane streaming reads "
                "(MI355X_MICROARCH.md, HBM section), hence fetch_bytes_corrected = 2*1024*FETCH_SIZE; WRITE_SIZE is exact for these stores",
       "kernels": {}}
for short, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    path = os.path.join(root, "gpurun_out", "pmc_%s_n1024_%s" % (tag, cname), "p_counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == cname:
            agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].strip()].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        d = out["kernels"].setdefault(k, {})
        d[cname + "_KiB_mean"] = sum(v) / len(v)
        d["launches_" + short] = len(v)
for k, d in out["kernels"].items():
    if "FETCH_SIZE_KiB_mean" in d:
        d["fetch_bytes_corrected"] = 2 * 1024 * d["FETCH_SIZE_KiB_mean"]
    if "WRITE_SIZE_KiB_mean" in d:
        d["write_bytes"] = 1024 * d["WRITE_SIZE_KiB_mean"]
jos = {e: v for k, v in out["kernels"].items() for e in (1, 2) if "gemm_f32_mfma_kernel<true, 1, %d>" % e in k}
if len(jos) == 2:
    per = {e: v["fetch_bytes_corrected"] + v["write_bytes"] for e, v in jos.items()}
    out["p_update_gemm_traffic_bytes_per_launch"] = 0.5 * (per[1] + per[2])
    out["first_joseph_gemm_traffic_bytes"] = per[1]
    out["second_joseph_gemm_traffic_bytes"] = per[2]
    # operands + C in + C out; the second GEMM (round 6) reads the lower triangle of T and writes both triangles
    out["first_joseph_gemm_algorithmic_bytes"] = 4.0 * (2 * n * m_pad + 2 * n * n)
    out["second_joseph_gemm_algorithmic_bytes"] = 4.0 * (2 * n * m_pad + 1.5 * n * n)
    out["p_update_gemm_algorithmic_bytes_per_launch"] = 0.5 * (out["first_joseph_gemm_algorithmic_bytes"] + out["second_joseph_gemm_algorithmic_bytes"])
dst = os.path.join(root, "profiles", "%s_pmc_traffic_n1024.json" % name)
json.dump(out, open(dst, "w"), indent=1)
print(dst)
for k, d in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("fetch_bytes_corrected", 0))[:10]:
    print("%-50s fetch %8.2f MB  write %8.2f MB" % (k[:50], d.get("fetch_bytes_corrected", 0) / 1e6, d.get("write_bytes", 0) / 1e6))
