"""Per-frame GPU timeline of the node's loop from a rocprofv3 kernel + memory-copy trace (scripts/frame_timeline.sh):
for the last frames of the run, every GPU activity with its start offset, duration and the idle gap in front of it."""
import csv, glob, os, sys
d = sys.argv[1]
ev = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:44]))
for path in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", ""))[:30]))
ev.sort()
# frames end with frame_outputs_kernel (or publish_status_kernel)
ends = [i for i, e in enumerate(ev) if e[2].startswith("frame_outputs_kernel") or e[2].startswith("publish_status_kernel")]
if len(ends) < 6:
    print("too few frames", len(ends)); sys.exit(0)
for fi in (len(ends) - 4, len(ends) - 3):
    a, b = ends[fi - 1] + 1, ends[fi]
    t0 = ev[a][0]
    print("frame: %d GPU activities, %.1f us from the first start to the last end" % (b - a + 1, (ev[b][1] - t0) / 1e3))
    prev_end = None
    busy = 0
    for s, e, n in ev[a:b + 1]:
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print("  +%7.1f us  %-46s %6.1f us   gap before %5.1f" % ((s - t0) / 1e3, n, (e - s) / 1e3, gap))
        prev_end = max(prev_end or e, e)
        busy += e - s
    print("  busy %.1f us" % (busy / 1e3))
