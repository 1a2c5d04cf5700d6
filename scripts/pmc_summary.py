"""Summarises the two rocprofv3 --pmc passes of scripts/gpu_pmc.sh into profiles/<tag>_pmc_traffic_n256.json.
Usage: python scripts/pmc_summary.py <tag> [<profile name>]   (reads gpurun_out/pmc_<tag>_{FETCH,WRITE}_SIZE/p_counter_collection.csv)"""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1]                                   # gpurun_out tag of the two passes
name = sys.argv[2] if len(sys.argv) > 2 else tag      # profiles/<name>_pmc_traffic_n256.json
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 40 "
                 "--warmup 10 --no-cpu-baseline, N=256, MI355X",
       "units": "FETCH_SIZE/WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts half of the bytes of 16-B/lane streaming "
                "reads (MI355X_MICROARCH.md, HBM section), hence fetch_bytes_corrected = 2*1024*FETCH_SIZE; WRITE_SIZE is exact "
                "for these stores",
       "kernels": {}}
for short, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    path = os.path.join(root, "gpurun_out", "pmc_%s_%s" % (tag, cname), "p_counter_collection.csv")
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
# round 6: ONE P-update GEMM per step (Sigma' = T2 + K G'^T, EPI 3) where the persistent launch forms T2; before: the pair (EPI 1, EPI 2)
one = [v for k, v in out["kernels"].items() if "gemm16_kernel<48, 2, 3>" in k and v.get("launches_fetch", 0) >= 40]
jos = [v for k, v in out["kernels"].items() if "gemm16_kernel<48, 2, 1>" in k or "gemm16_kernel<48, 2, 2>" in k]
if len(one) == 1:
    out["p_update_gemm_traffic_bytes_per_launch"] = one[0]["fetch_bytes_corrected"] + one[0]["write_bytes"]
    out["p_update_gemm_algorithmic_bytes_per_launch"] = 4 * (2 * 790 * 512 + 2 * 790 * 790)
    out["p_update_gemm"] = "gemm16_kernel<48, 2, 3>: Sigma' = T2 + K G'^T, one launch per step (T2 flow)"
elif len(jos) == 2:
    out["p_update_gemm_traffic_bytes_per_launch"] = 0.5 * sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in jos)
    out["p_update_gemm_algorithmic_bytes_per_launch"] = 4 * (2 * 790 * 512 + 2 * 790 * 790)
dst = os.path.join(root, "profiles", "%s_pmc_traffic_n256.json" % name)
json.dump(out, open(dst, "w"), indent=1)
print(dst)
for k, d in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("fetch_bytes_corrected", 0))[:8]:
    print("%-50s fetch %8.2f MB  write %8.2f MB" % (k[:50], d.get("fetch_bytes_corrected", 0) / 1e6, d.get("write_bytes", 0) / 1e6))
print("P-update GEMM traffic per launch: %.1f MB (algorithmic %.1f MB)" % (out.get("p_update_gemm_traffic_bytes_per_launch", 0) / 1e6, out.get("p_update_gemm_algorithmic_bytes_per_launch", 0) / 1e6))
