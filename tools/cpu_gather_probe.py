#!/usr/bin/env python3
"""What the host CPUs of the box can gather -- and what the process they are measured in does to the figure.  bench.py's cpu_baseline
leg runs inside a process that has torch (with its own bundled libgomp) and the HIP runtime loaded; round 3's all-core figures from
there (index_select on 128 threads slower than one core) were not credible.  This probe times the same OpenMP row gather
(oracle/coala_oracle.c orc_gather_rows_mt) and torch.index_select over a thread sweep in child processes of increasing baggage:

  clean      no torch in the process, table in ordinary memory
  torch      torch imported, GPU never touched, table in ordinary memory
  gpu        torch imported, GPU initialised, table in pinned host memory (hipHostMalloc) -- what bench.py's leg sees
  gpu+warm   the same after one parallel torch op has started torch's own OpenMP pool

  python tools/cpu_gather_probe.py [--rows 2000000 --dim 1024 --batch 28500]      (development tool; prints a table)
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(mode, rows, dim, batch):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
    import numpy as np
    torch = None
    if mode != "clean":
        import torch
    from oracle import oracle as O
    table = None
    keep = None
    if mode.startswith("gpu"):
        torch.cuda.init()
        torch.zeros(1, device="cuda")
        from COALA_GNN.synthetic import alloc_pinned_table
        keep = alloc_pinned_table(rows, dim, 0, 0)
        table = keep.array
        if mode == "gpu+warm":
            a = torch.randn(2048, 2048)
            (a @ a).sum().item()
    else:
        table = np.empty((rows, dim), dtype=np.float32)
        table[...] = 1.0
    rng = np.random.default_rng(0)
    idx = [rng.integers(0, rows, size=batch).astype(np.int64) for _ in range(12)]
    out = np.zeros((batch, dim), dtype=np.float32)
    cpus = len(os.sched_getaffinity(0))
    res = {"mode": mode, "cpus": cpus, "omp_gather_gbs": {}, "index_select_gbs": {}}
    for th in [t for t in (1, 2, 4, 8, 16, 32, 64, 128, 256) if t <= cpus]:
        O.gather_rows_mt(table, idx[0], out, th)
        t0 = time.perf_counter()
        for i in idx:
            O.gather_rows_mt(table, i, out, th)
        res["omp_gather_gbs"][th] = round(len(idx) * batch * dim * 4 / (time.perf_counter() - t0) / 1e9, 2)
        if th >= 8 and res["omp_gather_gbs"][th] < 0.3 * max(res["omp_gather_gbs"].values()):
            break
    if torch is not None:
        tt = torch.from_numpy(table)
        it = [torch.from_numpy(i) for i in idx]
        for th in [t for t in (1, 2, 4, 8, 16, 32, 64, 128, 256) if t <= cpus]:
            torch.set_num_threads(th)
            torch.index_select(tt, 0, it[0])
            t0 = time.perf_counter()
            for i in it:
                torch.index_select(tt, 0, i)
            res["index_select_gbs"][th] = round(len(idx) * batch * dim * 4 / (time.perf_counter() - t0) / 1e9, 2)
            if th >= 8 and res["index_select_gbs"][th] < 0.3 * max(res["index_select_gbs"].values()):
                break
    print(json.dumps(res), flush=True)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2_000_000)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=28500)
    ap.add_argument("--child", type=str, default=None)
    a = ap.parse_args()
    if a.child:
        return child(a.child, a.rows, a.dim, a.batch)
    print(f"# {a.rows} x {a.dim} fp32 table ({a.rows * a.dim * 4 / 1e9:.1f} GB), 12 batches of {a.batch} random rows per thread count; GB/s of rows gathered; "
          f"host CPUs {os.cpu_count()}, allowed {len(os.sched_getaffinity(0))}")
    for env_extra, label in (({}, ""), ({"OMP_WAIT_POLICY": "passive"}, " OMP_WAIT_POLICY=passive")):
        for mode in ("clean", "torch", "gpu", "gpu+warm"):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode, "--rows", str(a.rows), "--dim", str(a.dim), "--batch", str(a.batch)],
                                 capture_output=True, text=True, timeout=600, env=dict(os.environ, **env_extra))
            lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not lines:
                print(f"{mode}{label}: failed: {out.stderr[-300:]}")
                continue
            d = json.loads(lines[-1])
            print(f"{mode + label:34s} omp row gather  " + "  ".join(f"{k}:{v}" for k, v in d["omp_gather_gbs"].items()))
            if d["index_select_gbs"]:
                print(f"{'':34s} index_select    " + "  ".join(f"{k}:{v}" for k, v in d["index_select_gbs"].items()))


if __name__ == "__main__":
    main()
