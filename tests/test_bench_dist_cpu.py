"""world_size-2 gloo rehearsal of bench.py's multi-rank plumbing (replicas only: barrier,
max-over-ranks timing, whole-job aggregation).  No compute, no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_gloo_aggregation():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100",
           "--warmup", "5", "--selftest-dist"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["steps"] == 100
    # max over ranks of (0.5, 0.75) = 0.75 s; value = 2 sequences * 100 steps / 0.75 s
    assert abs(r["ms_per_step"] - 7.5) < 1e-6
    assert abs(r["value"] - 2 * 100 / 0.75) < 1e-6


def test_single_rank_selftest_line_has_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-dist", "--steps", "10"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in r
    assert r["vs_baseline"] is None and r["dtype"] == "f32" and "workload" in r["config"]


def test_gpus_flag_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no torchrun around it: the script starts its two ranks itself (before anything
    touches a GPU) and rank 0 reports the whole job."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100", "--warmup", "5",
                          "--selftest-dist"], env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and abs(r["value"] - 2 * 100 / 0.75) < 1e-6 and r["config"]["sequences"] == 2


def test_gpus_flag_that_disagrees_with_the_launcher_fails_loudly():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-dist"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_a_rank_that_dies_before_the_rendezvous_ends_the_run_with_its_stderr():
    """rank 1 dying at import / rendezvous used to leave rank 0 in the process group's barrier for its
    timeout with no diagnostic.  The self-spawning parent now polls every child, ends the others and shows the failed
    rank's stderr."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["EKFVIO_BENCH_SELFTEST_FAIL_RANK"] = "1"
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-dist"], env=env,
                         capture_output=True, text=True, timeout=150)
    assert out.returncode != 0 and time.time() - t0 < 120
    assert "rank 1 exited" in out.stderr and "dies before the rendezvous" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_eight_rank_rehearsal_over_gloo_under_the_drivers_launcher():
    """the driver's N = 8 launch line (torch.distributed.run, one rank per GPU) rehearsed on the CPU with
    the plumbing only.  The ranks meet over gloo -- bench.py never creates an RCCL group (BASELINE north_star: no RCCL) -- each
    rank's clock spans its own steps, and the max over ranks (rank 7: 0.5 + 7 * 0.25 s) is taken after the clocks stop."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", "29641", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "200",
           "--warmup", "10", "--selftest-dist"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["config"]["sequences"] == 8 and r["scaling"] == "weak"
    assert abs(r["value"] - 8 * 200 / 2.25) < 1e-6


def test_bench_never_asks_for_an_rccl_process_group():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"nccl"' not in src and "'nccl'" not in src
    assert 'backend="gloo"' in src
