#!/usr/bin/env python3
"""A multi-GPU configuration of BASELINE.json in its DISTRIBUTED form on ONE MI355X, timed: G logical ranks (host threads of this
process), the cold table pinned on the host and partitioned by owner (rank r pins rows r, r+G, ...), one shard of the partitioned
cache per rank, the configuration's fan-out at bs = 1024, sampler output bucketed by owner -> the product's fused native fetch
(coala_cache_fetch_distributed_bucketed) over the in-process transport.  Default = configs[4]: IGB-large 10,10,10, 16 GiB of cache
per rank + pinned-host spill, with the node count scaled to what the box's host memory holds (--rows; rows_scale_factor is reported:
the full table is 100 M x 1024 x 4 B = 409.6 GB; examples/ssd_gnn_dataloader.py:375-376).

What transfers to 8 GPUs: the hit ratio, the rows / bytes per step and per owner, K1's and K2's per-launch figures (each launch has
the GPU to itself here exactly as there).  What does not: the step time -- here 8 ranks share ONE PCIe link and ONE HBM, and the
exchange is a device-to-device copy instead of xGMI -- so `value` is "payload through one GPU playing all 8", reported as such.

Every `--check-every`-th step all delivered rows are compared bit for bit with the table's formula.
    python tools/dist_config_probe.py > profiles/r03_bench_igb_large_scaled.json"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "coala-gnn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import COALA_GNN_Pybind as P  # noqa: E402
from COALA_GNN.COALA_GNN_Manager import NativeExchange  # noqa: E402
from COALA_GNN.sampler import NeighborSampler  # noqa: E402
from COALA_GNN.synthetic import PinnedFeatureTable, feature_rows_torch, fill_table_partition, powerlaw_csc  # noqa: E402
from COALA_GNN_Pybind import _capi  # noqa: E402


def log(msg):
    print(f"[dist_config_probe] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rows", type=int, default=50_000_000)
    ap.add_argument("--full-rows", type=int, default=100_000_000, help="node count of the configuration itself (for rows_scale_factor)")
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--cache-mb", type=int, default=16384)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--fanout", type=str, default="10,10,10")
    ap.add_argument("--avg-degree", type=float, default=10.5)
    ap.add_argument("--prewarm", type=int, default=24)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--check-every", type=int, default=8)
    ap.add_argument("--name", type=str, default="BASELINE.json configs[4]: IGB-large GraphSAGE fan-out 10,10,10 bs=1024, 16 GB/GPU cache + pinned-host spill, 8 ranks")
    args = ap.parse_args()
    G, dim, fan = args.ranks, args.dim, [int(f) for f in args.fanout.split(",")]
    max_sample = args.batch * int(np.prod([f + 1 for f in fan]))
    torch.cuda.set_device(0)
    L = _capi.load()
    t0 = time.time()
    shards = []
    for r in range(G):
        t = PinnedFeatureTable((args.rows - r + G - 1) // G, dim, 0)
        fill_table_partition(t.cpu_tensor, 6, r, G, device="cuda:0")
        shards.append(t)
    log(f"{G} shards, {sum(t.nbytes for t in shards) / 1e9:.1f} GB pinned + filled in {time.time() - t0:.1f}s")
    ctrl = P.SSD_GNN_SSD_Controllers(1, dim * 4, 1024, 0, 0, dim, True)
    caches = [P.Isolated_Cache(ctrl, None, r, G, args.cache_mb, shards[r].device_ptr, num_rows=args.rows, rank=r, cold_partitioned=True,
                               sync=False, profile=True, max_batch=max_sample * 2) for r in range(G)]
    geo = caches[0].geometry()
    t0 = time.time()
    indptr, indices = powerlaw_csc(args.rows, args.avg_degree, seed=0, device="cuda")
    log(f"graph {args.rows} nodes / {indices.numel()} edges in {time.time() - t0:.1f}s")
    train = torch.randperm(int(0.6 * args.rows), generator=torch.Generator().manual_seed(1))
    spe = len(train) // (G * args.batch) - 1
    group = C.c_void_p()
    _capi.check(L.coala_comm_group_create(G, C.byref(group)))
    exs = [NativeExchange(None, 0, r, G, 0, inproc_group=group) for r in range(G)]
    samplers = [NeighborSampler(fan, seed=0, bucket_by_owner=G) for _ in range(G)]
    graphs = [s.make_graph(indptr, indices) for s in samplers]
    total = args.prewarm + args.steps
    res = [None] * G
    seen0 = [[None] * G for _ in range(total)]   # [step][source rank] -> the ids routed to owner 0
    errors = []
    bar = threading.Barrier(G, timeout=600)
    wall = [0.0]

    def worker(r):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            rows_n, checked, max_n = 0, 0, 0
            host = {"sample": 0.0, "alloc": 0.0, "fetch_call": 0.0, "check": 0.0, "record": 0.0}
            clk = time.perf_counter
            with torch.cuda.stream(stream):
                for step in range(total):
                    if step == args.prewarm:
                        stream.synchronize()
                        bar.wait()
                        caches[r].stats(reset=True)
                        caches[r].profile(reset=True)
                        bar.wait()
                        if r == 0:
                            wall[0] = time.perf_counter()
                    lo = ((step % spe) * G + r) * args.batch
                    t_a = clk()
                    ids, _, blocks = samplers[r].sample(graphs[r], train[lo: lo + args.batch].cuda(), step=step)
                    n = ids.numel()
                    assert args.batch <= n <= max_sample
                    t_b = clk()
                    feat = torch.empty((n, dim), dtype=torch.float32, device="cuda")
                    t_c = clk()
                    exs[r].fetch_bucketed(caches[r], feat.data_ptr(), ids.data_ptr(), n, blocks[0].owner_counts.data_ptr())
                    t_d = clk()
                    if step % args.check_every == args.check_every - 1:
                        for a in range(0, n, 1 << 15):
                            assert torch.equal(feat[a: a + (1 << 15)], feature_rows_torch(ids[a: a + (1 << 15)], dim, 6)), f"rank {r} step {step}: rows differ from the table"
                        checked += 1
                    t_e = clk()
                    # what owner 0 is asked for in this step by this rank: bucket 0 of the bucketed ids (replayed solo below)
                    n0 = int(exs[r].last_send_counts[0])
                    seen0[step][r] = ids[:n0].cpu().numpy()
                    t_f = clk()
                    if step >= args.prewarm:
                        for k, v in (("sample", t_b - t_a), ("alloc", t_c - t_b), ("fetch_call", t_d - t_c), ("check", t_e - t_d), ("record", t_f - t_e)):
                            host[k] += v
                    if step >= args.prewarm:
                        rows_n += n
                        max_n = max(max_n, n)
                    del feat
                    if r == 0 and step % 4 == 3:
                        stream.synchronize()
                        log(f"step {step + 1}/{total}")
                stream.synchronize()
                bar.wait()
            if r == 0:
                wall[0] = time.perf_counter() - wall[0]
            hit, miss, bad = caches[r].stats()
            p = caches[r].profile()
            res[r] = {"host_ms_per_step": {k: round(v / args.steps * 1e3, 3) for k, v in host.items()},
                      "hit": int(hit), "miss": int(miss), "bad": int(bad), "rows_requested": rows_n, "max_rows_in_a_minibatch": max_n, "rows_checked_bit_exact_steps": checked,
                      "k1_launches": int(p.gather_launches), "k1_ms": p.gather_ms, "k1_rows": int(p.gather_rows), "k1_hits": int(p.gather_hits),
                      "k2_launches": int(p.fill_launches), "k2_ms": p.fill_ms, "k2_rows": int(p.fill_rows)}
        except BaseException as e:  # noqa: BLE001
            errors.append((r, repr(e)))
            bar.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errors:
        raise SystemExit(f"failed: {errors}")
    # ---- owner 0's side of the same run, ALONE on the GPU: its one batch per step = the concatenation, in source-rank order, of what
    #      every rank routed to it.  A fresh shard-0 cache replays all steps through the owner-side serve (distributed set index, its
    #      own pinned shard), so every K1 / K2 launch has the GPU and the PCIe link to itself as on a real 8-GPU node; the hit / miss
    #      counters over the measured steps must equal the ones owner 0 kept in the 8-rank run (determinism contract).
    log("solo replay of owner 0")
    solo = P.Isolated_Cache(ctrl, None, 0, G, args.cache_mb, shards[0].device_ptr, num_rows=args.rows, rank=0, cold_partitioned=True,
                            sync=False, profile=True, max_batch=max_sample * 2)
    buf = None
    solo_rows = 0
    for step in range(total):
        if step == args.prewarm:
            torch.cuda.synchronize()
            solo.stats(reset=True)
            solo.profile(reset=True)
        batch0 = torch.from_numpy(np.concatenate(seen0[step])).cuda()
        if buf is None or buf.shape[0] < batch0.numel():
            buf = torch.empty((int(batch0.numel() * 1.2), dim), dtype=torch.float32, device="cuda")
        solo.serve(buf.data_ptr(), batch0.data_ptr(), batch0.numel())
        if step >= args.prewarm:
            solo_rows += batch0.numel()
    torch.cuda.synchronize()
    sh, sm, _ = solo.stats()
    sp = solo.profile()
    assert (int(sh), int(sm)) == (res[0]["hit"], res[0]["miss"]), f"owner 0 alone: hit/miss {(sh, sm)} differ from the 8-rank run's {(res[0]['hit'], res[0]['miss'])}"
    solo.close()
    del buf
    hit, miss = sum(x["hit"] for x in res), sum(x["miss"] for x in res)
    rows = sum(x["rows_requested"] for x in res)
    k1_l, k1_ms, k1_rows, k1_hits = int(sp.gather_launches), sp.gather_ms, int(sp.gather_rows), int(sp.gather_hits)
    k2_l, k2_ms, k2_rows = int(sp.fill_launches), sp.fill_ms, int(sp.fill_rows)
    tag_bytes = getattr(geo, "tag_set_bytes", 256) or 256
    k1_alg = k1_rows * (8 + tag_bytes) + k1_hits * 2 * dim * 4
    k1_us = k1_ms / max(k1_l, 1) * 1e3
    k2_us = k2_ms / max(k2_l, 1) * 1e3
    line = {
        "metric": "feature-gather GB/s (payload = rows x dim x 4 B delivered per second) -- ONE GPU playing all ranks: not a multi-GPU number",
        "value": round(rows * dim * 4 / wall[0] / 1e9, 3), "unit": "GB/s", "n_gpus": 1, "logical_ranks": G, "steps": args.steps, "warmup": args.prewarm,
        "ms_per_step": round(wall[0] / args.steps * 1e3, 3), "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.name, "rows": args.rows, "rows_scale_factor": round(args.rows / args.full_rows, 4), "dim": dim,
                   "table_GB": round(args.rows * dim * 4 / 1e9, 1), "cold_tier": f"pinned host (hipHostMalloc), owner-partitioned: {G} shards of {shards[0].nbytes / 1e9:.1f} GB",
                   "cache_mb_per_rank": args.cache_mb, "sets_per_rank": int(geo.num_sets), "line_bytes": int(geo.line_bytes), "fanout": args.fanout, "batch": args.batch,
                   "max_sample_rows": max_sample, "aggregate_cache_share_of_table": round(G * args.cache_mb * 2**20 / (args.rows * dim * 4), 3),
                   "transport": "in-process (host threads, device-to-device copies)", "input_nodes": "bucketed by owner by the sampler",
                   "hit_ratio": round(hit / max(hit + miss, 1), 4), "rows_per_step_per_rank": round(rows / G / args.steps, 1),
                   "max_rows_in_a_minibatch": max(x["max_rows_in_a_minibatch"] for x in res),
                   "cold_bytes_per_step_all_ranks_GB": round(miss / args.steps * dim * 4 / 1e9, 3),
                   "parity_check": f"all rows of every {args.check_every}th step == table formula, bit-exact ({sum(x['rows_checked_bit_exact_steps'] for x in res)} rank-steps)",
                   "steps_per_epoch_at_this_scale": spe},
        "roofline": {"bound": "hbm", "kernel": "probe_gather_kernel, owner 0's launches replayed ALONE on the GPU (one per step over everything routed to it)",
                     "achieved": round(k1_alg / max(k1_l, 1) / (k1_us * 1e-6) / 1e9, 1) if k1_us else None, "peak": 8000.0, "unit": "GB/s",
                     "frac": round(k1_alg / max(k1_l, 1) / (k1_us * 1e-6) / 1e9 / 8000.0, 4) if k1_us else None, "traffic": None,
                     "avg_launch_us": round(k1_us, 2), "launches": k1_l, "rows_per_launch": round(k1_rows / max(k1_l, 1), 1),
                     "hits_per_launch": round(k1_hits / max(k1_l, 1), 1), "alg_bytes_per_launch": int(k1_alg / max(k1_l, 1)),
                     "timing": "HIP events attached to each launch of the solo replay; hit/miss counters of the replay == owner 0's in the 8-rank run"},
        "roofline_cold_fill": {"bound": "pcie", "kernel": "miss_fill_kernel", "achieved": round(k2_rows * dim * 4 / max(k2_ms, 1e-9) / 1e6, 2), "peak": 63.0,
                               "unit": "GB/s", "frac": round(k2_rows * dim * 4 / max(k2_ms, 1e-9) / 1e6 / 63.0, 4), "avg_launch_us": round(k2_us, 1),
                               "launches": k2_l, "rows_per_launch": round(k2_rows / max(k2_l, 1), 1), "note": "owner 0's cold fill replayed alone: one launch per step over its whole batch"},
        "per_rank": res,
    }
    print(json.dumps(line), flush=True)
    for e in exs:
        e.close()
    _capi.check(L.coala_comm_group_destroy(group))
    for g in graphs:
        g.close()
    for c in caches:
        c.close()
    for t in shards:
        t.close()


if __name__ == "__main__":
    main()
