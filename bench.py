#!/usr/bin/env python3
"""bench.py — filter steps/s of the EKF-VIO hot path on MI355X.

One step = process(dt) + updateWithFeaturePositions with every landmark measured
(BASELINE.json metric).  Default workload at 1 GPU is config 2:
N = 256 landmarks (n = 790 states, m = 512 measurement rows), dt = 1/30, synthetic
closed-loop sequence, measurements resident in HBM before the timed region.
For --gpus N > 1 every rank runs an independent sequence on its own GPU (the filter is
sequential per step; the only shard is the sequence: replicas only, no collective on the
data path) and value = total steps/s over all ranks.  Launched under torchrun the ranks come
from the environment; launched plainly (`python bench.py --gpus N`) the script starts its N
ranks itself, before anything in the parent touches a GPU.

Prints ONE JSON line (rank 0).  Besides the contract keys: `roofline` (the P-update GEMM: one launch per step since round 6, a pair before),
`roofline_step` (the whole step), `cpu_baseline` (fp32 oracle, 1 core), `full_loop` (frames/s of
ekfvio_step_image: image in, KLT supplies z, update, replenishment), `klt` and `klt_cpu_baseline`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec peak


def dist_setup(n_gpus, backend="gloo"):
    """The ranks exchange one barrier and one float (the max-over-ranks time): always the gloo backend on 127.0.0.1.  No
    RCCL anywhere (BASELINE north_star: replicas, no collective on the data path) -- an RCCL barrier is a GPU kernel per
    rank, which has no business next to a 2 ms timed region."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, n_gpus):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch N ranks (torchrun) or none (the script then "
                         "starts them itself)" % (n_gpus, world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        import datetime
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    return world, rank, local


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: N child processes, one per device (RANK = LOCAL_RANK = r,
    seed = rank), rendezvous on 127.0.0.1.  The parent never initialises a GPU; rank 0's JSON line is the output."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        err = tempfile.TemporaryFile()
        errs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=err))
    # poll every child: a rank that dies before the rendezvous would otherwise leave the others (and this parent) in the
    # process group's barrier until its timeout.  The first failure ends the run and its stderr is shown.
    out0 = b""
    os.set_blocking(procs[0].stdout.fileno(), False)
    failed = None
    while True:
        try:
            chunk = procs[0].stdout.read()
        except BlockingIOError:
            chunk = None
        if chunk:
            out0 += chunk
        rcs = [p.poll() for p in procs]
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(rc == 0 for rc in rcs):
            break
        time.sleep(0.05)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        errs[failed].seek(0)
        sys.stderr.write("bench.py: rank %d exited with code %d; its stderr:\n%s\n" % (
            failed, procs[failed].returncode, errs[failed].read().decode(errors="replace")[-4000:]))
        return abs(procs[failed].returncode) or 1
    try:
        rest = procs[0].stdout.read()
        if rest:
            out0 += rest
    except BlockingIOError:
        pass
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return 0


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(x, world):
    if world == 1:
        return x
    import torch
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64)  # gloo: a host tensor
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def result_line(args, world, n_landmarks, elapsed, extra):
    """elapsed: the timed region's duration (max over ranks); with repetitions, their median."""
    n, m = 22 + 3 * n_landmarks, 2 * n_landmarks
    line = {
        "metric": "filter steps/sec (predict+update)", "value": world * args.steps / elapsed, "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "simLaunch-style closed loop: process(dt)+update per step, all landmarks measured, "
                               "measurements resident in HBM (BASELINE config 2)" if n_landmarks == 256 else
                               "synthetic closed loop, process(dt)+update per step",
                   "landmarks": n_landmarks, "state_dim": n, "measurement_rows": m, "dt": 1.0 / 30.0,
                   "sequences": world, "parallelism": "replicas (one independent sequence per GPU, no collective)",
                   "predict": args.predict},
    }
    line.update(extra)
    return line


def selftest_dist(args):
    """Exercises only the multi-rank plumbing (self-spawned or torchrun ranks, barrier, max-over-ranks, aggregation)
    with a synthetic per-rank time; used by the gloo CPU tests.  Computes nothing."""
    if os.environ.get("EKFVIO_BENCH_SELFTEST_FAIL_RANK") == os.environ.get("RANK", "0"):
        raise SystemExit("selftest: rank %s dies before the rendezvous" % os.environ.get("RANK", "0"))
    world, rank, _ = dist_setup(args.gpus)
    barrier(world)
    t0 = time.perf_counter()
    mine = 0.5 + 0.25 * rank + 0.0 * (time.perf_counter() - t0)  # (the per-rank clock of the real run, synthetic here)
    barrier(world)
    elapsed = max_over_ranks(mine, world)
    if rank == 0:
        print(json.dumps(result_line(args, world, args.landmarks, elapsed, {"selftest": True})))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def cpu_baseline(n_landmarks, dt, budget_steps, threads=1):
    """The oracle (line-faithful fp32 restatement of the reference filter) timed on the host: on ONE
    core (the reference is single-threaded: no threads/OpenMP in its CMakeLists) and, as a second
    figure, with the oracle's OpenMP products on all cores of this box's share."""
    from oracle import OracleFilter, set_threads
    from ekf_vio_amd.sim import Scenario
    set_threads(threads)
    sc = Scenario(n_landmarks, seed=0)
    f = OracleFilter(np.float32)
    f.add_new_features(sc.initial_features())
    fr = list(sc.frames(2 + budget_steps))
    for z, R, p in fr[:2]:
        f.process(dt), f.update(z, R, p)
    t0 = time.perf_counter()
    for z, R, p in fr[2:]:
        f.process(dt), f.update(z, R, p)
    el = time.perf_counter() - t0
    set_threads(1)
    return {"value": budget_steps / el, "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": "%d steps of process+update at N=%d after a 2-step warm-up, fp32 oracle "
                      "(oracle/ekf_oracle.hpp), %d thread%s; %.1f s" % (budget_steps, n_landmarks, threads,
                                                                        "" if threads == 1 else "s (OpenMP)", el)}


# ---------------------------------------------------------------------------------------------------------------
# Full loop ("KLT reported separately and inside the full-loop figure"): ekfvio_step_image per frame
# = frame upload, pyramid, process(dt), KLT seeded by the filter's prediction, update, FAST replenishment.
def _textured_sequence(frames, dx=-1.4, dy=-0.45):
    """The reference's 640x480 test image seen by a camera gliding over it (ekf_vio_amd.sim.translated_sequence)."""
    from PIL import Image
    from ekf_vio_amd.sim import translated_sequence
    base = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "images", "640_480_test_gray.png")))
    return translated_sequence(base, frames, dx, dy)


def full_loop(n_landmarks, device, frames=40, warm=6, node_defaults=False, outputs=False):
    from ekf_vio_amd import EKFVIO, capi
    K = np.array([500.0, 0, 320.0, 0, 500.0, 240.0, 0, 0, 1.0], np.float32)
    prof = 8
    imgs = _textured_sequence(warm + frames + prof)
    # enough corners for N landmarks on this image: the reference defaults (threshold 50, 30 px apart) yield ~90
    thr, dist = (50, 30) if n_landmarks <= 64 else (20, 12)
    extra = {}
    if node_defaults:  # the node's own parameter defaults (Params.h): the frame is resized by 4 before anything else
        extra["inverse_image_scale"] = 4
    v = EKFVIO(max_features=n_landmarks, device=device, replenish=1, fast_threshold=thr, min_new_feature_dist=dist, **extra)
    e = v.tc_ekf
    numeric = 0
    for i in range(warm):
        numeric += v.addFrame(i / 30.0, imgs[i], K) == capi.ENUMERIC
    e.synchronize()
    n_start = e.num_features
    # four chunks with a synchronise behind each; the rate is that of the MEDIAN chunk (a 40-frame region is 4-6 ms of host-driven calls: one
    # scheduler or allocator hiccup of a millisecond in it moved a whole-region figure by 25 % between two visits); the whole region is reported beside it
    chunk_s, nchunk = [], 4
    t0 = time.perf_counter()
    for c in range(nchunk):
        tc = time.perf_counter()
        for i in range(warm + c * frames // nchunk, warm + (c + 1) * frames // nchunk):
            numeric += v.addFrame(i / 30.0, imgs[i], K) == capi.ENUMERIC
            if outputs:  # what the node publishes per frame (EKFVIO.cpp:444-518): odometry and the landmark cloud
                v.odometry()
                v.points()
        e.synchronize()
        chunk_s.append((time.perf_counter() - tc) / max(1, (c + 1) * frames // nchunk - c * frames // nchunk))
    el_whole = time.perf_counter() - t0
    el = float(np.median(chunk_s)) * frames
    st = e.get_state()
    tracked = int((st["del_flag"] == 0).sum())
    # per-stage device time of the same loop, HIP events on the handle's stream (adds a host wait per stage: not timed above)
    e.profile(True)
    for i in range(warm + frames, warm + frames + prof):  # the sequence simply continues
        v.addFrame(i / 30.0, imgs[i], K)
    rep = e.profile_report()
    e.profile(False)
    stage = {k: 1e3 * x["ms"] / prof for k, x in rep.items() if x["launches"]}
    finite = bool(np.isfinite(st["base_mu"]).all() and np.isfinite(st["Sigma"]).all())
    N = e.num_features
    e.close()
    pyr_us, trk_us = stage.get("klt_pyramid", 0.0), stage.get("klt_track", 0.0)
    # algorithmic bytes of the pyramid build per frame: read the frame, write levels 0-3 (8-bit) and
    # their int16 (dx, dy) derivatives
    lv = [(640, 480), (320, 240), (160, 120), (80, 60)]
    pyr_bytes = 640 * 480 + sum(w * h * (1 + 4) for w, h in lv)
    return {"frames_per_s": frames / el, "ms_per_frame": 1e3 * el / frames, "frames_per_s_whole_region": frames / el_whole, "landmarks": N, "landmarks_at_start": n_start,
            "landmarks_never_lost": tracked, "numeric_warnings": int(numeric), "state_finite": finite, "frames": frames,
            "image": "tests/golden/images/640_480_test_gray.png translated by (-1.4, -0.45) px per frame, fx = fy = 500",
            "what": "ekfvio_step_image per frame: H2D frame, %spyramid, process(dt), KLT (z from the tracker), update, replenishment (cfg.replenish=1, FAST threshold %d, %d px apart)"
                    % ("resize by 4 inside the " if node_defaults else "", thr, dist)
                    + ("; then the node's per-frame outputs: odometry (ekfvio_get_odometry) and the point cloud with intensities (ekfvio_get_points)" if outputs else ""),
            "stage_us_per_frame": stage,
            "klt": {"pyramid_us": pyr_us, "track_us": trk_us, "tracks_per_s": (N / (trk_us * 1e-6)) if trk_us else None,
                    "pyramid_bytes": pyr_bytes,
                    "pyramid_gb_per_s": (pyr_bytes / (pyr_us * 1e-6) / 1e9) if pyr_us else None,
                    "pyramid_frac_of_hbm_peak": (pyr_bytes / (pyr_us * 1e-6) / 1e9 / PEAK_HBM_GBS) if pyr_us else None,
                    "note": "event-bracketed stage times (include ~5 us of launch gaps per stage); the pyramid is one launch over 0.3 MB in / 2.6 MB out whose duration is a workgroup's "
                            "latency, not HBM bound; the tracker is one wavefront per landmark, latency bound on its Gauss-Newton chain"}}


def klt_cpu_baseline(n_points):
    """the oracle's pyramidal LK (oracle/klt_oracle.cpp, OpenCV 3.x calcOpticalFlowPyrLK restated), 1 core:
    pyramids + Scharr derivatives of both frames, then n_points tracks from one frame of the sequence into the next."""
    from oracle import KltFrame, klt_track, set_threads
    set_threads(1)
    a, b = _textured_sequence(2)
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(40, 600, n_points), rng.uniform(40, 440, n_points)], axis=1).astype(np.float32)
    t0 = time.perf_counter()
    A, B = KltFrame(a), KltFrame(b)
    t1 = time.perf_counter()
    reps = 0
    while True:
        nxt, st, _ = klt_track(A, B, pts, pts.copy())
        reps += 1
        if time.perf_counter() - t1 > 2.0 or reps >= 20:
            break
    t2 = time.perf_counter()
    return {"pyramid_ms_per_frame": 1e3 * (t1 - t0) / 2, "track_ms": 1e3 * (t2 - t1) / reps, "points": n_points,
            "tracks_per_s": n_points * reps / (t2 - t1), "tracked_ok": int(st.sum()), "cores": 1, "kind": "port",
            "sample": "2 pyramids + %d x %d tracks, window 21, levels 0-3, 30 it / 0.01 (KLTTracker.cpp:61-64)" % (reps, n_points)}


def concurrent_sequences(n_landmarks, device, sequences, steps, dt_warm=10):
    """Several independent sequences on ONE GPU, each its own handle and stream, their graph replays in flight together:
    what the idle compute units are worth when a single filter step is one workgroup's latency.  (Config 4 itself is one
    sequence per GPU; this is reported next to it, never as `value`.)"""
    from ekf_vio_amd import TightlyCoupledEKF
    from ekf_vio_amd.sim import Scenario
    hs = []
    for s in range(sequences):
        sc = Scenario(n_landmarks, seed=100 + s)
        g = TightlyCoupledEKF(max_features=n_landmarks, device=device)
        g.addNewFeatures(sc.initial_features())
        fr = list(sc.frames(dt_warm + steps))
        g.upload_measurements(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]))
        g.run_uploaded(0, 0, sc.dt)
        hs.append((g, sc))
    for g, sc in hs:
        g.run_uploaded(0, dt_warm, sc.dt)
    for g, sc in hs:
        g.synchronize()
    for g, sc in hs:
        g.run_uploaded(dt_warm, 0, sc.dt)
    t0 = time.perf_counter()
    for g, sc in hs:
        g.run_uploaded(dt_warm, steps, sc.dt)
    for g, sc in hs:
        g.synchronize()
    el = time.perf_counter() - t0
    ok = all(bool(np.isfinite(g.base_mu).all()) for g, _ in hs)
    for g, _ in hs:
        g.close()
    return {"sequences": sequences, "steps_each": steps, "total_steps_per_s": sequences * steps / el,
            "per_sequence_steps_per_s": steps / el, "states_finite": ok}


def device_resident_rate(n_landmarks, device, steps=200, warm=20, predict="structured", depth_var=100.0, with_roofline=False):
    """steps/s of a device-resident run at another size or predict mode (an extra, never `value`): same procedure as the
    headline (graphs prepared first, `steps` steps between two synchronises), median of five readings."""
    from ekf_vio_amd import TightlyCoupledEKF, capi
    from ekf_vio_amd.sim import Scenario
    sc = Scenario(n_landmarks, seed=0)
    mode = capi.PREDICT_DENSE if predict == "dense" else capi.PREDICT_STRUCTURED
    g = TightlyCoupledEKF(max_features=n_landmarks, device=device, predict_mode=mode, default_point_depth_variance=depth_var)
    g.addNewFeatures(sc.initial_features())
    fr = list(sc.frames(warm + steps))
    g.upload_measurements(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]))
    g.run_uploaded(0, 0, sc.dt)
    g.run_uploaded(0, warm, sc.dt)
    g.synchronize()
    snap = g.get_state()
    times, warn = [], 0
    for _ in range(5):
        g.set_state(snap)
        g.run_uploaded(warm, 0, sc.dt)
        g.synchronize()
        t0 = time.perf_counter()
        g.run_uploaded(warm, steps, sc.dt)
        warn += g.synchronize() == capi.ENUMERIC
        times.append(time.perf_counter() - t0)
    out = {"landmarks": n_landmarks, "state_dim": 22 + 3 * n_landmarks, "steps": steps, "predict": predict,
           "steps_per_s": steps / float(np.median(times)), "ms_per_step": 1e3 * float(np.median(times)) / steps,
           "numeric_warnings": int(warn), "state_finite": bool(np.isfinite(g.base_mu).all()),
           "sweep": "persistent launch" if g.sweep_counts()["persistent"] else "one launch per block step"}
    if with_roofline:
        # BASELINE config 3 (the MFMA / HBM roofline stress size): the P-update GEMM pair measured live like the headline's (HIP
        # events on the handle's stream around 20 pairs replayed from a graph) and the per-stage device times of the step
        n, m_pad = 22 + 3 * n_landmarks, ((2 * n_landmarks + 63) // 64) * 64
        avg_us, flops_per_launch = g.profile_update_gemms(20)
        tf = flops_per_launch / (avg_us * 1e-6) / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": "P-update GEMM pair (Sigma - K W: all tiles; T + G K^T: the lower triangle's tiles, mirrored -- round 6): gemm_f32_mfma_kernel, 64 x 64 tiles, three workgroups per compute unit; flops_per_launch = EXECUTED flops, averaged over the pair",
                           "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F32_MFMA_TFLOPS,
                           "flops_per_launch": flops_per_launch, "avg_launch_us": avg_us, "shape": {"M": n, "N": n, "K": m_pad},
                           "traffic": None, "algorithmic_bytes_per_launch": 4.0 * (2.0 * n * m_pad + 2.0 * n * n),
                           "note": "pairs replayed back to back from a graph on the filter's own operands; rocprofv3 means of the same launches inside the step: profiles/r06_kernel_stats_n1024.csv"}
        for tag in ("r05", "r06"):  # (the newest record wins)
            pmc = os.path.join(ROOT, "profiles", "%s_pmc_traffic_n1024.json" % tag)
            if n_landmarks == 1024 and os.path.exists(pmc):
                pj = json.load(open(pmc))
                if "p_update_gemm_traffic_bytes_per_launch" in pj:
                    out["roofline"]["traffic"] = pj["p_update_gemm_traffic_bytes_per_launch"]
                    if "p_update_gemm_algorithmic_bytes_per_launch" in pj:
                        out["roofline"]["algorithmic_bytes_per_launch"] = pj["p_update_gemm_algorithmic_bytes_per_launch"]
                    out["roofline"]["traffic_quoted_from_profiles"] = True
                    out["roofline"]["traffic_source"] = "profiles/%s_pmc_traffic_n1024.json" % tag
        g.profile(True)
        g.run_uploaded(warm, 6, sc.dt)
        g.synchronize()
        rep = g.profile_report()
        g.profile(False)
        out["stage_us_per_step"] = {k: 1e3 * v["ms"] / 6 for k, v in rep.items() if v["launches"]}
        fl = step_flops(n_landmarks)
        if not g.counters()["t2_updates"] and os.environ.get("EKFVIO_SYM_JOSEPH", "1") != "0" and (n + 63) // 64 * ((n + 63) // 64) > 256:
            tn = (n + 63) // 64  # executed, not dense-form, flops: the second Joseph GEMM forms tn (tn + 1) / 2 of its tn^2 tiles
            cut = 2.0 * n * n * m_pad - 0.5 * tn * (tn + 1) * 2.0 * 64 * 64 * m_pad
            fl["joseph_gemms"] -= cut
            fl["total"] -= cut
        out["roofline_step"] = {"flops_per_step": fl["total"], "achieved": fl["total"] / (out["ms_per_step"] * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                                "unit": "TFLOP/s", "frac": fl["total"] / (out["ms_per_step"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}
    if predict == "dense":
        # the two dense F P F^T GEMMs (north-star form): event-bracketed device time of the gemm_predict class over 10 steps
        g.profile(True)
        g.run_uploaded(warm, 10, sc.dt)
        g.synchronize()
        rep = g.profile_report()
        g.profile(False)
        gp = rep.get("gemm_predict")
        if gp and gp["launches"]:
            n = 22 + 3 * n_landmarks
            us = 1e3 * gp["ms"] / gp["launches"]  # one scope = the X = F P and X F^T launches, back to back
            tf = 4.0 * n ** 3 / (us * 1e-6) / 1e12
            out["fp_gemm_pair"] = {"flops": 4.0 * n ** 3, "pair_us_event_bracketed": us, "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS,
                                   "unit": "TFLOP/s", "frac": tf / PEAK_F32_MFMA_TFLOPS,
                                   "note": "dense-formulation flops: F is [[A,0],[B,D]] with 358+36N non-zeros, so all but %.1f %% of these "
                                           "multiply structural zeros (SURVEY 8(d) honesty clause); eager launches, includes host launch gaps"
                                           % (100.0 * (358 + 36 * n_landmarks) / float(n * n))}
    g.close()
    return out


def step_flops(N, t2_flow=False):
    """Algorithmic flops the step executes (dense form of the update, structured predict), DESIGN.md section 4.  t2_flow (round 6, where the fused
    persistent launch forms the gain): T2 = Sigma - Y S Y^T inside that launch as tile pairs (the product is symmetric: n^2 m), the gain tiles'
    own copy of T2's measured rows (2 n m^2), and ONE Joseph GEMM behind the launch instead of two."""
    n, m = 22 + 3 * N, 2 * N
    m_pad = ((m + 63) // 64) * 64
    if t2_flow:
        joseph = 2.0 * n * n * m_pad                               # Sigma' = T2 + K G'^T
        in_launch = 1.0 * n * n * m_pad + 2.0 * n * m_pad * m_pad  # T2's tile pairs; (H T2)^T in the gain tiles
    else:
        joseph = 2.0 * n * (n + 1) * m_pad + 2.0 * n * n * m_pad   # T = Sigma - K W (+ the K*y column); Sigma' = T + G K^T
        in_launch = 0.0
    gain = 1.0 * n * m_pad * m_pad                               # K = Y L^-1, triangular operand
    sweep = m_pad ** 3 / 3.0 + (n + m_pad / 2.0) * m_pad * m_pad  # Cholesky + the two augmented row blocks
    predict = 4.0 * n * (358.0 + 36.0 * N)
    return {"joseph_gemms": joseph, "gain_gemm": gain, "t2_in_launch": in_launch, "cholesky_sweep": sweep, "structured_predict": predict,
            "total": joseph + gain + in_launch + sweep + predict,
            "north_star_dense_form": 4.0 * n ** 3 + m ** 3 / 3.0 + 4.0 * n * m * m + 4.0 * n * n * m}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--landmarks", type=int, default=256)
    ap.add_argument("--predict", choices=["structured", "dense"], default="structured")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-loop", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=0, help="oracle steps to time (0 = auto ~10-30 s)")
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--replay-only", action="store_true",
                    help="for rocprofv3 --kernel-trace --stats: nothing but warm-up and the timed region's graph replays runs (no eager per-stage pass, no "
                         "per-call loop, no GEMM micro-replay), so the per-kernel means are those of replayed steps and sum to ms_per_step")
    ap.add_argument("--selftest-dist", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))  # before any GPU call in this process
    if args.selftest_dist:
        return selftest_dist(args)

    import torch
    from ekf_vio_amd import TightlyCoupledEKF, capi
    from ekf_vio_amd.sim import Scenario
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    world, rank, local = dist_setup(args.gpus)
    torch.cuda.set_device(local)
    N = args.landmarks
    sc = Scenario(N, seed=rank)
    mode = capi.PREDICT_DENSE if args.predict == "dense" else capi.PREDICT_STRUCTURED
    g = TightlyCoupledEKF(max_features=N, device=local, predict_mode=mode)
    g.addNewFeatures(sc.initial_features())
    total = args.warmup + args.steps
    extra_frames = args.profile_steps + 100  # the profiled steps and the per-call (PCIe-inclusive) loop continue the sequence
    n_frames = min(total, 4096) + extra_frames  # longer runs wrap around the uploaded sequence
    fr = list(sc.frames(n_frames))
    g.upload_measurements(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]))
    dt = sc.dt
    g.run_uploaded(0, 0, dt)  # captures the launch graphs (nothing runs)
    g.run_uploaded(0, args.warmup, dt)
    g.synchronize()
    # an odd warm-up ends with one eager step, which flips the mean / covariance ping-pong the graphs were captured
    # for: prepare again (count = 0 recaptures if needed) so that no capture ever lands inside the timed region
    g.run_uploaded(args.warmup, 0, dt)
    g.synchronize()
    # The timed region is EXACTLY args.steps steps between barrier + synchronize on both sides.  When it is shorter than
    # 50 ms (20 steps are 2.5 ms) a single reading is at the mercy of one scheduling hiccup: the region is then repeated
    # on the SAME steps from the SAME state (restored outside the timed region) and `value` is the median repetition.
    snap = g.get_state()
    times, numeric = [], capi.OK

    def timed_region():
        # barrier + synchronize on both sides; every rank's own clock runs over its K steps only (enqueue, device work, the
        # handle's and the device's synchronise): the barrier in front lines the ranks up BEFORE t0, the one behind and the
        # max-over-ranks exchange come AFTER the clock has stopped
        torch.cuda.synchronize()
        barrier(world)
        t0 = time.perf_counter()
        g.run_uploaded(args.warmup, args.steps, dt)
        rc = g.synchronize()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        barrier(world)
        return max_over_ranks(mine, world), rc

    el, numeric = timed_region()
    times.append(el)
    repeats = 1
    if el < 0.050:
        repeats = 9
    repeats = int(max_over_ranks(float(repeats), world))  # every rank repeats as often as the slowest decision says
    for _ in range(repeats - 1):
        g.set_state(snap)
        g.run_uploaded(args.warmup, 0, dt)  # graphs re-prepared for the restored ping-pong orientation: no capture in the region
        g.synchronize()
        el, rc = timed_region()
        times.append(el)
        if rc == capi.ENUMERIC:
            numeric = rc
    elapsed = float(np.median(times))
    md, ma = g.checkSigma()
    st_ok = bool(np.isfinite(g.base_mu).all() and md >= 0)

    extra = {"state_finite_and_psd_diag": st_ok, "numeric_warning_in_timed_run": bool(numeric == capi.ENUMERIC),
             "timed_region": {"repeats": repeats, "value_is": "median" if repeats > 1 else "single reading",
                              "steps_per_s_min": world * args.steps / max(times), "steps_per_s_max": world * args.steps / min(times),
                              "region_ms": [1e3 * t for t in times],
                              "note": "each repetition times exactly --steps steps from the same restored state, barrier + synchronize on both sides; ranks meet over gloo "
                                      "(no RCCL), each rank's clock spans its own steps only, max over ranks taken after the clocks stop"},
             "parity": "oracle unpinned against the reference binary (the reference holds one KAT, test/test_ekf.cpp:44-63; "
                       "no Eigen/ROS/OpenCV in the image to build it): HIP vs own fp32/fp64 CPU restatement, tests/ -m gpu",
             "cpu_baseline_eigen_sparse": "unavailable: no Eigen3 on this box (BASELINE.md B3)"}
    if rank == 0 and args.replay_only:
        extra["replay_only"] = True
        extra["t2_updates"] = g.counters()["t2_updates"]
    elif rank == 0:
        # per-kernel-class device time with HIP events on the handle's stream
        g.profile(True)
        g.run_uploaded(total, args.profile_steps, dt)
        g.synchronize()
        rep = g.profile_report()
        g.profile(False)
        n, m_pad = 22 + 3 * N, ((2 * N + 63) // 64) * 64
        # The dominant arithmetic kernel: the two P-update GEMMs (T = Sigma - K W with the K*y column,
        # Sigma' = T + G K^T).  Their mean launch duration is measured live with HIP events on the handle's
        # stream around 50 pairs replayed back to back from a hipGraph, i.e. under the launch conditions of
        # the timed region (the per-class event brackets above run eagerly and include host launch gaps).
        avg_us, flops_per_launch = g.profile_update_gemms(50)
        achieved = flops_per_launch / (avg_us * 1e-6) / 1e12
        t2_flow = g.counters()["t2_updates"] > 0  # round 6: ONE P-update GEMM behind the persistent launch, which forms T2 = Sigma (I - K H)^T itself
        extra["roofline"] = {"bound": "mfma", "kernel": ("the P-update GEMM (Sigma' = T2 + K G'^T; T2 and G' come out of the persistent launch), tile kernel chosen by shape" if t2_flow
                                                         else "P-update GEMM pair (Sigma - K W, T + G K^T), tile kernel chosen by shape"),
                             "p_update_gemm_launches_per_step": 1 if t2_flow else 2,
                             "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                             "flops_per_launch": flops_per_launch, "avg_launch_us": avg_us,
                             "shape": {"M": n, "N": n, "K": m_pad},
                             "note": ("the product alone, replayed on scratch; inside the timed region's graphs the same launch also carries the NEXT step's "
                                      "linearisation in 33 workgroups of its own (profiles/r06_linearisation_overlap.txt; rocprofv3 mean of the launch there: "
                                      "profiles/r06_kernel_stats_n256.csv)") if t2_flow else "pairs replayed back to back on scratch"}
        # HBM-side traffic and MFMA-busy counters of the same kernels come from separate rocprofv3 --pmc passes
        # (bench.py cannot collect PMCs itself); the committed summaries are quoted when the workload matches
        for tag in (("r06",) if t2_flow else ("r05", "r04", "r03", "r02", "r01")):
            pmc = os.path.join(ROOT, "profiles", "%s_pmc_traffic_n256.json" % tag)
            if N == 256 and os.path.exists(pmc):
                pj = json.load(open(pmc))
                if "p_update_gemm_traffic_bytes_per_launch" in pj:
                    extra["roofline"]["traffic"] = pj["p_update_gemm_traffic_bytes_per_launch"]
                    extra["roofline"]["traffic_quoted_from_profiles"] = True  # not measured in this run
                    extra["roofline"]["traffic_source"] = "profiles/%s_pmc_traffic_n256.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 correction applied)" % tag
                    extra["roofline"]["algorithmic_bytes_per_launch"] = pj["p_update_gemm_algorithmic_bytes_per_launch"]
                    break
        for tag in (("r06",) if t2_flow else ("r05", "r04", "r03", "r02")):
            mf = os.path.join(ROOT, "profiles", "%s_pmc_mfma_n256.json" % tag)
            if N == 256 and os.path.exists(mf):
                mj = json.load(open(mf))
                extra["roofline"]["mfma_counters"] = mj.get("p_update_gemms")
                extra["roofline"]["mfma_counters_quoted_from_profiles"] = True  # not measured in this run
                extra["roofline"]["mfma_counters_source"] = "profiles/%s_pmc_mfma_n256.json (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ...)" % tag
                break
        # The whole step against the same peak: what fraction of the chip's fp32 matrix rate one filter step uses
        fl = step_flops(N, t2_flow)
        ms_step = 1e3 * elapsed / args.steps
        extra["roofline_step"] = {"bound": "mfma", "flops_per_step": fl["total"], "breakdown": {k: fl[k] for k in ("joseph_gemms", "gain_gemm", "t2_in_launch", "cholesky_sweep", "structured_predict")},
                                  "achieved": fl["total"] / (ms_step * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                  "frac": fl["total"] / (ms_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                  "north_star_dense_form_flops": fl["north_star_dense_form"],
                                  "note": "executed algorithmic flops (dense-form update on the padded shape, structured predict) / ms_per_step; the north-star dense form "
                                          "(4n^3 predict + Joseph expansion) is listed for reference and is NOT what runs: the predict exploits F's sparsity"}
        # PCIe-inclusive rate (never `value`): the per-call boundary with host-resident measurements,
        # one ekfvio_process + one synchronising ekfvio_update per step
        nh = 100
        import gc
        gc.collect()
        gc.disable()  # (with torch imported a generation-2 collection is a 40-70 ms pause: seen as one outlier call in some runs)
        th = time.perf_counter()
        per_call = []
        for z_h, R_h, p_h in fr[-nh:]:
            t1 = time.perf_counter()
            g.process(dt)
            g.updateWithFeaturePositions(z_h, R_h, p_h)
            per_call.append(time.perf_counter() - t1)
        g.synchronize()
        extra["pcie_inclusive_steps_per_s"] = nh / (time.perf_counter() - th)
        gc.enable()
        extra["pcie_inclusive_us_per_step"] = {"median": 1e6 * float(np.median(per_call)), "p90": 1e6 * float(np.percentile(per_call, 90)),
                                               "max": 1e6 * float(np.max(per_call)), "argmax": int(np.argmax(per_call)), "sweeps": g.sweep_counts()}
        # The same GEMM kernel family at the N=1024 stress shape (3094 x 3094 x 2048), where a launch is many rounds of
        # workgroups instead of one: what the kernel reaches when the shape lets it (the N=256 figure above is bounded
        # by one workgroup's latency plus the kernel boundary)
        try:
            import ctypes as C
            us = C.c_double(0)
            gh = TightlyCoupledEKF(max_features=4, device=local, hooks=True)  # (a test hook: lives in the hooks build of the library, not in the product)
            rc_h = gh.lib.ekfvio_test_gemm_bench(gh.h, 1, 0, 3094, 3094, 2048, 20, 0, C.byref(us))
            gh.close()
            if rc_h == 0 and us.value > 0:
                tf = 2.0 * 3094 * 3094 * 2048 / (us.value * 1e-6) / 1e12
                extra["roofline_stress_shape"] = {"shape": {"M": 3094, "N": 3094, "K": 2048}, "avg_launch_us": us.value,
                                                  "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                                  "frac": tf / PEAK_F32_MFMA_TFLOPS, "note": "plain C = A*B^T of the N=1024 Joseph shape, 20 launches"}
        except Exception as ex:  # diagnostic extra, never fatal
            extra["roofline_stress_shape"] = {"error": str(ex)}
        extra["stage_us_per_step"] = {k: 1e3 * v["ms"] / args.profile_steps for k, v in rep.items() if v["launches"]}
        if "gather" in extra["stage_us_per_step"] and g.sweep_counts()["persistent"]:
            # with the fused persistent launch the "gather" scope holds only the update's bookkeeping kernel (the measurement gather itself
            # is part of the persistent launch, scope "cholesky"): say so in the key
            extra["stage_us_per_step"]["update_bookkeeping"] = extra["stage_us_per_step"].pop("gather")
        # The sweep is the longest launch of the step (half of it at N=256), and it is NOT the roofline kernel: its duration is
        # the latency of one workgroup's sequential pivot chain, stated here so that nobody has to infer it
        if "cholesky" in extra["stage_us_per_step"]:
            us_sweep = extra["stage_us_per_step"]["cholesky"]
            extra["roofline_sweep"] = {"bound": "latency (sequential pivot chain of one workgroup; DESIGN.md section 3)",
                                       "kernel": "gather + Cholesky sweep + gain tiles in one persistent launch (chol_persist_kernel where it applies, else gather, one launch per block step, gain kernel)",
                                       "flops_per_step": fl["cholesky_sweep"] + fl["gain_gemm"] + fl["t2_in_launch"],
                                       "flops_breakdown": {"cholesky_sweep": fl["cholesky_sweep"], "gain": fl["gain_gemm"], "t2_and_its_measured_rows": fl["t2_in_launch"]},
                                       "stage_us": us_sweep,
                                       "achieved": (fl["cholesky_sweep"] + fl["gain_gemm"] + fl["t2_in_launch"]) / (us_sweep * 1e-6) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                       "frac": (fl["cholesky_sweep"] + fl["gain_gemm"] + fl["t2_in_launch"]) / (us_sweep * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                       "note": "event-bracketed stage time of the eager, per-stage timed run (includes a few us of launch gaps); "
                                               "since round 4 the stage is ONE launch that also holds the measurement gather, the first diagonal tile and the gain's tiles, "
                                               "since round 6 also T2 = Sigma (I - K H)^T, G' and K y (chol_persist_kernel, fused); in-kernel stamps: profiles/r06_persistent_t2.txt"}
    g.close()
    if rank == 0 and world == 1 and not args.no_full_loop and not args.replay_only:
        try:
            extra["concurrent_sequences_one_gpu"] = [concurrent_sequences(N, local, b, min(args.steps, 200)) for b in (2, 4, 8)]
        except Exception as ex:  # an extra, never fatal
            extra["concurrent_sequences_one_gpu"] = {"error": repr(ex)}
    if rank == 0:
        if world == 1 and not args.no_full_loop:
            try:  # extras, never fatal: the author's deployment sizes (params/test.yaml 30, the node default 100, fast_with_insight.yaml 400
                  # with its depth variance 1000) and the north-star dense predict beside the structured one (report both)
                extra["other_sizes"] = [device_resident_rate(30, local, depth_var=1000.0), device_resident_rate(100, local),
                                        device_resident_rate(400, local, depth_var=1000.0)]
                # BASELINE config 3 (N = 1024: n = 3094, m = 2048), driver-timed, with the roofline of its P-update GEMMs
                extra["config3_n1024"] = device_resident_rate(1024, local, steps=30, warm=6, with_roofline=True)
                extra["predict_dense"] = device_resident_rate(N, local, steps=100, predict="dense")
                extra["predict_dense"]["structured_steps_per_s_same_run"] = world * args.steps / elapsed
                # the sweep as one launch per block step (EKFVIO_SWEEP=0: what a node that shares its GPU runs, what several handles on one
                # device run, and what an aborted persistent launch is rerun with); the switch is read when the handle is created
                prev_sweep = os.environ.get("EKFVIO_SWEEP")
                os.environ["EKFVIO_SWEEP"] = "0"
                try:
                    extra["per_step_sweep_n256"] = device_resident_rate(N, local)
                finally:
                    if prev_sweep is None:
                        del os.environ["EKFVIO_SWEEP"]
                    else:
                        os.environ["EKFVIO_SWEEP"] = prev_sweep
                # round 6's same-run A/B: the two-GEMM flow of rounds 1-5 (EKFVIO_T2=0: gain inside the persistent launch, T = Sigma - K W and
                # Sigma' = T + G K^T as two GEMMs behind it) against the default T2 flow timed as `value` above
                prev_t2 = os.environ.get("EKFVIO_T2")
                os.environ["EKFVIO_T2"] = "0"
                try:
                    extra["two_gemm_flow_n256"] = device_resident_rate(N, local)
                finally:
                    if prev_t2 is None:
                        del os.environ["EKFVIO_T2"]
                    else:
                        os.environ["EKFVIO_T2"] = prev_t2
            except Exception as ex:
                extra["other_sizes"] = {"error": repr(ex)}
            try:
                extra["full_loop"] = {"n64": full_loop(64, local), "n256": full_loop(256, local),
                                      "n256_with_outputs": full_loop(256, local, outputs=True),
                                      "node_defaults_n100_scale4": full_loop(100, local, node_defaults=True),
                                      "node_defaults_with_outputs": full_loop(100, local, node_defaults=True, outputs=True)}
                extra["klt"] = extra["full_loop"]["n256"].pop("klt")
                extra["full_loop"]["n64"].pop("klt")
                extra["full_loop"]["node_defaults_n100_scale4"].pop("klt")
                extra["full_loop"]["node_defaults_with_outputs"].pop("klt")
            except Exception as ex:  # reported, never fatal for the headline metric
                extra["full_loop"] = {"error": repr(ex)}
        if world == 1 and not args.no_cpu_baseline:
            steps = args.cpu_steps or max(3, int(round(60.0 * (256.0 / N) ** 3)))
            extra["cpu_baseline"] = cpu_baseline(N, dt, steps)
            ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            if ncpu > 1:
                extra["cpu_baseline_all_cores"] = cpu_baseline(N, dt, max(3, steps // 2), threads=min(ncpu, 64))
            extra["klt_cpu_baseline"] = klt_cpu_baseline(256)
        print(json.dumps(result_line(args, world, N, elapsed, extra)))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
