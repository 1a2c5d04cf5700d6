"""One filter step's GPU timeline from a rocprofv3 kernel trace of bench.py (scripts/step_timeline.sh): every kernel of a
step in the middle of the run with its start offset, duration and the idle gap in front of it; medians over the steps."""
import csv, glob, os, sys, statistics
d = sys.argv[1]
ev = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:48]))
ev.sort()
starts = [i for i, e in enumerate(ev) if e[2].startswith("predict_fused_kernel")]
if len(starts) < 40:
    print("too few steps", len(starts)); sys.exit(0)
# steps from the middle of the run (the timed graph replays; the bench's last steps are its eager, per-stage timed ones)
mid = len(starts) // 2
steps = [(starts[i], starts[i + 1]) for i in range(mid - 8, mid + 8)]
import collections
n = collections.Counter(b - a for a, b in steps).most_common(1)[0][0]
steps = [s for s in steps if s[1] - s[0] == n]
per = statistics.median((ev[b][0] - ev[a][0]) / 1e3 for a, b in steps)
print("%d kernels per step, median step period %.2f us over %d steps" % (n, per, len(steps)))
for q in range(n):
    dur = statistics.median((ev[a + q][1] - ev[a + q][0]) / 1e3 for a, b in steps)
    off = statistics.median((ev[a + q][0] - ev[a][0]) / 1e3 for a, b in steps)
    gap = statistics.median((ev[a + q][0] - max(e[1] for e in ev[a + q - 1:a + q])) / 1e3 for a, b in steps) if q else statistics.median((ev[a][0] - ev[a - 1][1]) / 1e3 for a, b in steps)
    print("  +%7.2f us  %-50s %7.2f us   gap before %5.2f" % (off, ev[steps[0][0] + q][2], dur, gap))
