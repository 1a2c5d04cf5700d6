"""Eight sequences on one GPU under rocprofv3 --kernel-trace: where does the time of a step go when every handle takes the per-step sweep?
Usage on the GPU box:  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o c -- python3 $ROOT/scripts/concurrent_prof.py run
                       python3 scripts/concurrent_prof.py report OUT/c_kernel_trace.csv"""
import collections, csv, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if sys.argv[1] == "run":
    import numpy as np
    from ekf_vio_amd import TightlyCoupledEKF
    from ekf_vio_amd.sim import Scenario
    S, steps, warm, N = 8, 64, 10, 256
    hs = []
    for s in range(S):
        sc = Scenario(N, seed=100 + s)
        g = TightlyCoupledEKF(max_features=N)
        g.addNewFeatures(sc.initial_features())
        fr = list(sc.frames(warm + steps))
        g.upload_measurements(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]))
        g.run_uploaded(0, 0, sc.dt)
        hs.append((g, sc))
    for g, sc in hs:
        g.run_uploaded(0, warm, sc.dt)
    for g, sc in hs:
        g.synchronize()
    t0 = time.perf_counter()
    for g, sc in hs:
        g.run_uploaded(warm, steps, sc.dt)
    for g, sc in hs:
        g.synchronize()
    el = time.perf_counter() - t0
    print("8 sequences: %.0f steps/s in total" % (S * steps / el))
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the timed region = the last 8 * 64 steps: take the last 60 % of the trace
    t_end = max(int(r["End_Timestamp"]) for r in rows)
    t_beg = t_end - int(0.5 * (t_end - int(rows[0]["Start_Timestamp"])))
    sel = [r for r in rows if int(r["Start_Timestamp"]) >= t_beg]
    span = (t_end - t_beg) / 1e3
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in sel:
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:44]
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in agg.values())
    print("window %.0f us, kernel time summed over streams %.0f us (mean concurrency %.2f)" % (span, tot, tot / span))
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-46s n %5d  mean %7.1f us  share of summed time %.3f" % (k, n, t / n, t / tot))
    # how many kernels are in flight at once (time-weighted)
    ev = []
    for r in sel:
        ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
    ev.sort()
    cur, last, hist = 0, ev[0][0], collections.Counter()
    for t, d in ev:
        hist[cur] += t - last
        last, cur = t, cur + d
    T = sum(hist.values())
    print("kernels in flight (share of the window):", {k: round(v / T, 3) for k, v in sorted(hist.items())})
