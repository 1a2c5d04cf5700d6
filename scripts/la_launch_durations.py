"""Per-launch durations of chol_step_la_kernel in the last full step of a rocprofv3 kernel trace: python scripts/la_launch_durations.py <kernel_trace.csv>"""
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "chol_step_la" in r["Kernel_Name"]]
last = rows[-31:]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
print("workgroups", " ".join(str(int(r["Grid_Size_X"]) // 256) for r in last))
print("us        ", " ".join("%.1f" % x for x in d), " sum %.1f" % sum(d))
