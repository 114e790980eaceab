"""bench.py contract (the driver's entry point): one JSON line with the required keys, at N=1 and -- through the one-GPU
development hook -- under torch.distributed.run with 2 ranks, including the end-to-end epoch leg."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--rows", "300000", "--prewarm", "20", "--steps", "10", "--warmup", "3", "--cache-mb", "256", "--epoch-steps", "12",
         "--cpu-baseline-batches", "2", "--allhit-launches", "5", "--no-color-affinity-leg"]
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _line(out):
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:]
    return json.loads(lines[0])


def test_bench_single_gpu_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL], capture_output=True, text=True, timeout=900)
    d = _line(out)
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 3 and d["value"] > 0
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and "workload" in d["config"]
    r = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r) and r["bound"] == "hbm" and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(c) and c["kind"] == "port" and c["value"] > 0
    assert r["traffic_source"] is None          # not the profiled default command line -> no PMC constant is quoted
    e = d["epoch"]
    assert e["serial"]["steps"] == 12 and e["prefetch"] is None and e["serial"]["ms_per_step"] > 0   # one loader in the line (--epoch-prefetch adds the other)
    x = d["config_fanout_10_10"]                # the 10,10 batch shape has an N = 1 origin too
    assert x["value"] > 0 and x["steps"] == 60 and x["rows_per_step_per_gpu"] > d["config"]["rows_per_step_per_gpu"]


def test_bench_two_ranks_on_one_gpu_with_epoch_leg():
    env = dict(os.environ, COALA_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL, "--epoch-prefetch-multi"]
    d = _line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env))
    assert d["n_gpus"] == 2 and d["config"]["cache_backend"] == "nccl" and d["config"]["TEST_HOOK_single_device"] is True
    assert d["cpu_baseline"] is None and d["value"] > 0
    e = d["epoch"]
    assert "error" not in e and e["serial"]["steps"] == 12 and e["prefetch"]["steps"] == 12
    assert "ONE all-reduce of a flat buffer" in e["model"]      # the harness's own gradient averaging (the default at N > 1)
    xg = d["exchange"]
    assert xg["bound"] == "xgmi" and xg["avg_us"] > 0 and xg["remote_bytes_in_per_gpu_per_step"] > 0 and xg["peak"] == 153.0
    x = d["config_fanout_10_10"]
    assert "error" not in x and x["value"] > 0 and x["steps"] == 60 and 0 <= x["hit_ratio"] <= 1
    assert d["config"]["exchange_transport"] == "torch" and "parity_check" in d["config"]   # gloo hook: no RCCL group -> torch transport
    rp = d["config"]["exchange_rounds"]         # measured during the warm-up, the same choice on every rank
    assert rp["chosen"] in (1, 2, 4) and set(rp["ms_per_fetch"]) == {"1", "2", "4"} and all(v > 0 for v in rp["ms_per_fetch"].values())


def test_bench_self_launch_two_ranks():
    """`python3 bench.py --gpus 2` with NO launcher around it (the form a driver uses at N = 1): bench.py starts its two ranks as child
    processes before anything touches the GPU, relays rank 0's line and returns the worst exit code (examples/4GB_script.sh:28-37: one
    process per GPU, started by the launcher)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["COALA_BENCH_SINGLE_DEVICE"] = "1"
    small = [a for a in SMALL if a != "--no-color-affinity-leg"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *small, "--no-fanout-leg", "--affinity-nodes", "150000",
                          "--affinity-cache-mb", "16", "--affinity-steps", "40"], capture_output=True, text=True, timeout=900, env=env)
    d = _line(out)
    assert d["n_gpus"] == 2 and d["value"] > 0 and "error" not in d and d["config"]["TEST_HOOK_single_device"] is True
    assert d["epoch"]["serial"]["steps"] == 12
    assert "launcher: rank exit codes [0, 0]" in out.stderr
    # f-3 on the ranks of the job (node_distributor_pybind.cuh:150-222): 2 domains x 1 rank, real colours, both modes bit-exact and exactly partitioned
    ca = d["color_affinity_2x1"]
    assert ca["num_colors"] > 10 and ca["baseline"]["steps"] == 40 and ca["node_color"]["steps"] == 40
    for mode in ("baseline", "node_color"):
        assert ca[mode]["rows_bit_exact_steps_per_rank"] == 3 and ca[mode]["global_batches_partitioned_exactly"] is True
        assert len(ca[mode]["per_domain"]) == 2 and all(0 < x["hit_ratio"] < 1 and x["ranks"] == 1 for x in ca[mode]["per_domain"])
    assert ca["node_color"]["hit_ratio_all_domains"] > ca["baseline"]["hit_ratio_all_domains"] - 0.01


def test_bench_stall_ends_inside_the_time_budget():
    """A rank that stops making progress (what a first-contact hang of a transport looks like): every watchdog deadline is cut off at the
    run's time budget, so the job prints its line with an "error" field and leaves non-zero well before a driver's own limit."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(COALA_BENCH_SINGLE_DEVICE="1", COALA_BENCH_INJECT_STALL="1:first_fetch")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL, "--time-budget", "75"], capture_output=True,
                         text=True, timeout=600, env=env)
    took = time.time() - t0
    assert out.returncode != 0 and took < 75 + 45, (out.returncode, took)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads(lines[0])
    assert "abandoned by the watchdog" in d["error"] and "time budget" in d["error"]


def test_bench_extra_legs_are_skipped_when_the_budget_is_short():
    """With little of the time budget left the extra legs are skipped and say so; the headline line still comes out with code 0."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, "--time-budget", "74", "--no-cpu-baseline", "--no-allhit"],
                         capture_output=True, text=True, timeout=600)
    d = _line(out)
    assert d["value"] > 0 and "skipped" in d["epoch"] and "error" not in d


def test_bench_epoch_leg_stops_at_its_soft_limit_on_every_rank():
    """An epoch leg that is merely slow must not end as a watchdog failure: at its soft time limit -- one decision for all ranks, taken on the
    host every 64 steps -- the leg stops, reports what it measured as an extrapolation labelled `cut_short`, and the run goes on (rc 0)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(COALA_BENCH_SINGLE_DEVICE="1", COALA_BENCH_SOFT_LIMIT_S="0.01")
    small = [a for a in SMALL]
    small[small.index("--epoch-steps") + 1] = "-1"        # a whole epoch (86 steps per rank at this size): cut at the first check, step 64
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *small, "--no-fanout-leg"], capture_output=True, text=True,
                         timeout=600, env=env)
    d = _line(out)
    e = d["epoch"]["serial"]
    assert e["steps"] == 64 and "cut_short" in e and e["epoch_time_s_extrapolated"] > 0 and "error" not in d


def test_bench_fallback_to_torch_transport_two_ranks():
    """The first-minibatch check fails on ONE rank (injected): every rank must agree (flag over the CPU group), drop its exchange and go
    on over the torch transport, and the line says so (COALA_GNN_Manager.py:159-203 is what the fall-back re-creates in Python)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(COALA_BENCH_SINGLE_DEVICE="1", COALA_BENCH_INJECT_EXCHANGE_FAIL="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL, "--no-fanout-leg", "--epoch-steps", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    d = _line(out)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["exchange_transport"].startswith("torch (fallback")
    assert d["config"]["input_nodes"] == "sampler order"


def test_bench_fallback_from_native_rccl_exchange_one_rank():
    """The same fall-back from the NATIVE exchange on real RCCL objects (one-rank rehearsal): the native exchange runs its first minibatch,
    is declared failed (injected), its RCCL communicator is destroyed, and the run continues over torch's communicator."""
    env = dict(os.environ, COALA_BENCH_INJECT_EXCHANGE_FAIL="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, "--backend", "nccl", "--no-fanout-leg", "--epoch-steps", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    d = _line(out)
    c = d["config"]
    assert c["rccl_rehearsal_one_rank"] is True and c["exchange_transport"].startswith("torch (fallback: the native exchange")
    assert d["value"] > 0 and c["counts_ahead"] is False


def test_bench_one_rank_rccl_rehearsal():
    """`bench.py --backend nccl` at N = 1: a torch.distributed world of one rank with RCCL for GPU tensors, the fused native exchange on
    its OWN one-rank RCCL communicator (ranks as ncclCommCount reports them), sampler-bucketed ids, count exchanges issued ahead, and
    the epoch leg under DistributedDataParallel (torch's RCCL communicator next to the exchange's) -- what the multi-GPU run does,
    minus bytes on a link.  Runs on the driver's one-GPU box every round."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, "--backend", "nccl", "--no-fanout-leg", "--epoch-prefetch", "--ddp"], capture_output=True,
                         text=True, timeout=900)
    d = _line(out)
    c = d["config"]
    assert d["n_gpus"] == 1 and c["cache_backend"] == "nccl" and c["rccl_rehearsal_one_rank"] is True
    assert c["exchange_transport"] == "native" and c["rccl_ranks"] == 1 and c["counts_ahead"] is True
    assert "bucketed by owner" in c["input_nodes"] and d["value"] > 0 and 0 < d["roofline"]["frac"] < 1
    e = d["epoch"]
    assert e["model"].endswith("DistributedDataParallel") and e["serial"]["steps"] == 12 and e["prefetch"]["steps"] == 12   # (--ddp: torch's wrapper)


def test_bench_rank_failure_is_visible():
    """A rank-local exception inside an extra leg must not strand the other ranks nor look like a success: rank 0 prints the
    line with an "error" field and the job ends with a non-zero exit code, quickly."""
    import time
    env = dict(os.environ, COALA_BENCH_SINGLE_DEVICE="1", COALA_BENCH_INJECT_FAIL="1:serial")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL, "--no-fanout-leg"]
    t0 = time.time()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0, "a failed rank was reported as success"
    assert time.time() - t0 < 400
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads(lines[0])
    assert "injected failure on rank 1" in d["error"] and d["value"] > 0 and "error" in d["epoch"]


def test_color_affinity_probe_two_domains():
    """f-3: two domains x 1 rank on GPU 0, real colours from the native colouring tool on a planted-community graph; node_color and
    baseline must both deliver the table's rows bit-exact and partition every global batch exactly; the hit ratios are reported."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "color_affinity_probe.py"), "--nodes", "150000", "--dim", "128",
                          "--cache-mb", "8", "--batch", "256", "--max-steps", "60", "--community", "512"],
                         capture_output=True, text=True, timeout=600)
    d = _line(out)
    for mode in ("baseline", "node_color"):
        assert len(d[mode]) == 2
        for r in d[mode]:
            assert r["steps"] == 60 and r["rows_verified_bit_exact_steps"] == 5 and r["global_batches_partitioned_exactly"] is True
            assert 0 < r["hit_ratio"] < 1 and r["fetch_ms_per_step"] > 0
    assert d["num_colors"] > 10
    h = d["hit_ratio_all_domains"]
    assert h["node_color"] > h["baseline"] - 0.01, h     # on a graph with communities affinity routing must not lose


def test_backend_compare_probe_logical_ranks():
    """examples/Cache_compare_script.sh on one GPU: isolated caches vs the owner-partitioned cache with 3 logical ranks (host threads, the
    fused native fetch over the in-process transport).  Both modes deliver the table's rows bit-exact; sharing one sharded cache can
    only raise the aggregate hit ratio at equal per-rank capacity."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "backend_compare_probe.py"), "--ranks", "3", "--rows", "300000", "--dim", "256",
                          "--cache-mb", "16", "--batch", "256", "--steps", "120", "--measure-from", "60"], capture_output=True, text=True, timeout=600)
    d = _line(out)
    for mode in ("isolated", "partitioned"):
        assert len(d[mode]["per_rank"]) == 3 and all(r["rows_checked_bit_exact_steps"] >= 1 for r in d[mode]["per_rank"])
        assert 0 < d[mode]["hit_ratio_all_ranks"] < 1
    assert d["partitioned"]["hit_ratio_all_ranks"] > d["isolated"]["hit_ratio_all_ranks"]
