"""Summarises the rocprofv3 --pmc pass of scripts/gpu_pmc_mfma.sh into profiles/<name>_pmc_mfma_n256.json.
Usage: python scripts/pmc_mfma_summary.py <tag> [<profile name>]
MFMA utilisation of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles): the counter sums, over
all SIMDs of the chip, the cycles in which a SIMD's matrix pipe was busy (MI355X_MICROARCH.md: "counts cycles").  Kernel
cycles come from the kernel-trace duration x the clock GRBM_GUI_ACTIVE implies (sum over the 8 XCDs / 8 / duration)."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else tag
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
d = os.path.join(root, "gpurun_out", "pmc_%s_mfma" % tag)
cc = list(csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))))
kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(os.path.join(d, "p_kernel_trace.csv")))}
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in cc:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].strip()
    per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    t = kt.get(r["Dispatch_Id"])
    if t is not None and r["Counter_Name"] == "SQ_BUSY_CYCLES":
        per[k]["duration_ns"].append(float(t["End_Timestamp"]) - float(t["Start_Timestamp"]))
out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE "
                 "--kernel-trace -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-full-loop, N=256, MI355X",
       "notes": "means per launch.  mfma_busy_frac_of_simd_time = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x clock x duration), "
                "clock = GRBM_GUI_ACTIVE / 8 / duration (the guide's effective-clock formula; reads high on launches this short, "
                "so the fraction is a lower bound).  mfma_mops_f32 counts fp32 MFMA ops in units of 512 flops (MOPS) where the counter is populated.",
       "kernels": {}}
for k, c in per.items():
    e = {n: sum(v) / len(v) for n, v in c.items()}
    e["launches"] = len(c.get("SQ_BUSY_CYCLES", []))
    dur = e.get("duration_ns")
    if dur and e.get("GRBM_GUI_ACTIVE") and e.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        clk_ghz = e["GRBM_GUI_ACTIVE"] / 8.0 / dur
        e["effective_clock_ghz"] = clk_ghz
        e["mfma_busy_frac_of_simd_time"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * clk_ghz * dur)
        e["mfma_busy_frac_at_2p4ghz"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * 2.4 * dur)
    out["kernels"][k] = e
gem = {k: v for k, v in out["kernels"].items() if any(("gemm16_kernel<48, 2, %d>" % e) in k for e in (1, 2, 3)) and v.get("launches", 0) >= 40}
if gem:
    out["p_update_gemms"] = {k: {n: v.get(n) for n in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "duration_ns",
                                                        "effective_clock_ghz", "mfma_busy_frac_of_simd_time", "mfma_busy_frac_at_2p4ghz")}
                             for k, v in gem.items()}
dst = os.path.join(root, "profiles", "%s_pmc_mfma_n256.json" % name)
json.dump(out, open(dst, "w"), indent=1)
print(dst)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:8]:
    print("%-44s mfma_busy %12.0f  dur %8.0f ns  frac %s" % (k[:44], v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), v.get("duration_ns", 0),
                                                            "%.3f" % v["mfma_busy_frac_of_simd_time"] if "mfma_busy_frac_of_simd_time" in v else "n/a"))
