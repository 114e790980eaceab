#!/usr/bin/env python3
"""Isolated per-GPU caches against the owner-partitioned cache at equal per-rank capacity, on ONE GPU.

The reference's cache comparison (examples/Cache_compare_script.sh:28-34: `isolated` vs `nccl` / `nvshmem` backends) needs N GPUs.
What it measures, though -- G data-parallel ranks either caching what each of them touches (duplicates across ranks) or sharing
one cache sharded by id % G (no duplicates, G x the distinct lines) -- can be reproduced with G logical ranks on one device: here
the G ranks are G host threads of this process, each with its own HIP stream, sampler handle and cache handle, and the
partitioned mode runs the product's fused native fetch (coala_cache_fetch_distributed_bucketed) over the in-process transport.
Hit ratios transfer to N GPUs as they are (the cache algorithm sees the same batches); times do not (one PCIe link and one HBM
are shared by all ranks here, and the exchange is a device copy instead of xGMI) and are reported only per mode.

  python tools/backend_compare_probe.py [--ranks 4 --rows 4000000 --cache-mb 1600 --steps 600]"""
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
import torch  # noqa: E402
import COALA_GNN_Pybind as P  # noqa: E402
from COALA_GNN.COALA_GNN_Manager import NativeExchange  # noqa: E402
from COALA_GNN.sampler import NeighborSampler  # noqa: E402
from COALA_GNN.synthetic import alloc_pinned_table, feature_rows_torch, powerlaw_csc  # noqa: E402
from COALA_GNN_Pybind import _capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=4)
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--cache-mb", type=int, default=1600, help="per rank (default: 10 %% of the table, the ratio of BASELINE configs[1])")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--fanout", type=str, default="5,5")
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--measure-from", type=int, default=300, help="steps before this one warm the caches")
    ap.add_argument("--counts-ahead", action="store_true", help="partitioned mode: the count exchange of step t+1 goes out (sampler stream) before "
                                                                 "the fetch of step t, which then has no host synchronisation")
    args = ap.parse_args()
    G, dim, fan = args.ranks, args.dim, [int(f) for f in args.fanout.split(",")]
    torch.cuda.set_device(0)
    L = _capi.load()
    table = alloc_pinned_table(args.rows, dim, 0, 0)
    indptr, indices = powerlaw_csc(args.rows, 12.0, seed=0, device="cuda")
    train = torch.randperm(int(0.6 * args.rows), generator=torch.Generator().manual_seed(0))
    ctrl = P.SSD_GNN_SSD_Controllers(1, dim * 4, 1024, 0, 0, dim, True)
    spe = len(train) // (G * args.batch)          # steps per epoch of the data-parallel run
    out = {}
    for mode in ("isolated", "partitioned"):
        part = mode == "partitioned"
        caches = [P.Isolated_Cache(ctrl, None, r, G if part else 1, args.cache_mb, table.device_ptr, num_rows=args.rows, rank=r if part else 0, sync=False,
                                   max_batch=args.batch * (fan[0] + 1) * (fan[1] + 1) * (G if part else 1)) for r in range(G)]
        group = C.c_void_p()
        exs = None
        if part:
            _capi.check(L.coala_comm_group_create(G, C.byref(group)))
            exs = [NativeExchange(None, 0, r, G, 0, inproc_group=group) for r in range(G)]
        samplers = [NeighborSampler(fan, seed=0, bucket_by_owner=G if part else 0) for _ in range(G)]
        graphs = [s.make_graph(indptr, indices) for s in samplers]
        res = [None] * G
        errors = []
        bar = threading.Barrier(G, timeout=300)

        def worker(r):
            try:
                torch.cuda.set_device(0)
                stream = torch.cuda.Stream()
                side = torch.cuda.Stream()
                gpu_ms, rows_n, checked = 0.0, 0, 0

                def sample(step):
                    lo = ((step % spe) * G + r) * args.batch  # rank r takes the r-th batch of the global batch
                    seeds = train[lo: lo + args.batch].cuda()
                    if not (part and args.counts_ahead):
                        return samplers[r].sample(graphs[r], seeds, step=step), None
                    with torch.cuda.stream(side):
                        smp = samplers[r].sample(graphs[r], seeds, step=step)
                        tk = exs[r].counts_begin(smp[2][0].owner_counts.data_ptr())
                        ev = torch.cuda.Event()
                        ev.record()
                    torch.cuda.current_stream().wait_event(ev)
                    return smp, tk

                with torch.cuda.stream(stream):
                    nxt = sample(0)
                    for step in range(args.steps):
                        if step == args.measure_from:
                            stream.synchronize()
                            bar.wait()
                            caches[r].stats(reset=True)
                            bar.wait()
                            t0 = time.perf_counter()
                        (ids, _, blocks), tk = nxt
                        if step + 1 < args.steps:
                            nxt = sample(step + 1)
                        n = ids.numel()
                        feat = torch.empty((n, dim), dtype=torch.float32, device="cuda")
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record()
                        if part:
                            exs[r].fetch_bucketed(caches[r], feat.data_ptr(), ids.data_ptr(), n, blocks[0].owner_counts.data_ptr(), ticket=tk)
                        else:
                            caches[r].read_feature(feat.data_ptr(), ids.data_ptr(), n)
                        b.record()
                        if step % 150 == 7:
                            assert torch.equal(feat, feature_rows_torch(ids, dim, 0)), f"{mode}: rank {r} step {step}: delivered rows differ from the table"
                            checked += 1
                        if step >= args.measure_from:
                            b.synchronize()
                            gpu_ms += a.elapsed_time(b)
                            rows_n += n
                    stream.synchronize()
                    bar.wait()
                wall = time.perf_counter() - t0
                hit, miss, bad = caches[r].stats()
                res[r] = {"hit": int(hit), "miss": int(miss), "fetch_ms_per_step_on_stream": round(gpu_ms / (args.steps - args.measure_from), 4),
                          "wall_ms_per_step": round(wall / (args.steps - args.measure_from) * 1e3, 4), "rows_per_step": round(rows_n / (args.steps - args.measure_from), 1),
                          "rows_checked_bit_exact_steps": checked}
            except BaseException as e:  # noqa: BLE001
                errors.append((r, repr(e)))
                bar.abort()

        ts = [threading.Thread(target=worker, args=(r,)) for r in range(G)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errors:
            raise SystemExit(f"{mode}: {errors}")
        # in the partitioned mode a rank's counters are those of the OWNER (what it served for everybody): the sum is what matters
        hit, miss = sum(x["hit"] for x in res), sum(x["miss"] for x in res)
        out[mode] = {"per_rank": res, "hit_ratio_all_ranks": round(hit / max(hit + miss, 1), 4), "misses_per_step_all_ranks": round(miss / (args.steps - args.measure_from), 1),
                     "cold_bytes_per_step_all_ranks_MB": round(miss / (args.steps - args.measure_from) * dim * 4 / 1e6, 2)}
        if part:
            for e in exs:
                e.close()
            _capi.check(L.coala_comm_group_destroy(group))
        for g in graphs:
            g.close()
        for c in caches:
            c.close()
    line = {"what": f"isolated caches vs the owner-partitioned cache, {G} logical ranks (host threads) on one MI355X, {args.cache_mb} MB per rank, "
                    f"measured over steps {args.measure_from}..{args.steps} of a data-parallel run (global batch = {G} x {args.batch})",
            "rows": args.rows, "dim": dim, "fanout": args.fanout, "table_GB": round(args.rows * dim * 4 / 1e9, 2),
            "aggregate_cache_share_of_table": round(G * args.cache_mb * 2**20 / (args.rows * dim * 4), 3), **out,
            "note": "hit ratios transfer to N GPUs unchanged; times do not (all ranks share one PCIe link / HBM here and the exchange is a device copy)"}
    print(json.dumps(line), flush=True)
    table.close()


if __name__ == "__main__":
    main()
