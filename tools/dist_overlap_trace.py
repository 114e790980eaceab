#!/usr/bin/env python3
"""Evidence for "the row exchange runs beside the cold fill on a second HIP stream" on ONE GPU: G logical ranks (host threads) run
the product's native distributed fetch (coala_cache_fetch_distributed_bucketed, in-process transport: the rows of a round are
device-to-device copies on the communicator's own stream) on a workload with a real miss ratio, under
`rocprofv3 --kernel-trace` (tools/dist_overlap_profile.sh).  `--summarize <kernel_trace.csv>` then reads the trace: for every
cold-fill launch, which row copies started while it was running.  Development tool.

  python tools/dist_overlap_trace.py [--ranks 2 --steps 12 --rounds 4]"""
import argparse
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))


def run(args):
    import ctypes as C
    import threading
    import torch
    import COALA_GNN_Pybind as P
    from COALA_GNN.COALA_GNN_Manager import NativeExchange
    from COALA_GNN.synthetic import PinnedFeatureTable, feature_rows_torch, fill_table_partition
    from COALA_GNN_Pybind import _capi
    torch.cuda.set_device(0)
    L = _capi.load()
    G, dim, rows, n = args.ranks, 1024, 2_000_000, 28500
    shards = []
    for r in range(G):
        t = PinnedFeatureTable((rows - r + G - 1) // G, dim, 0)
        fill_table_partition(t.cpu_tensor, 0, r, G, device="cuda:0")
        shards.append(t)
    ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
    caches = [P.Isolated_Cache(ctrl, None, r, G, 1024, shards[r].device_ptr, num_rows=rows, rank=r, cold_partitioned=True, sync=False, max_batch=2 * n)
              for r in range(G)]
    group = C.c_void_p()
    _capi.check(L.coala_comm_group_create(G, C.byref(group)))
    exs = [NativeExchange(None, 0, r, G, 0, inproc_group=group, rounds=args.rounds) for r in range(G)]
    hot = torch.randperm(rows, generator=torch.Generator().manual_seed(0))[: 6 * n]      # a working set a little larger than the caches keep hot
    errors = []
    bar = threading.Barrier(G, timeout=120)

    def worker(r):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            gen = torch.Generator().manual_seed(10 + r)
            with torch.cuda.stream(stream):
                for step in range(args.steps):
                    mix = torch.cat([hot[torch.randperm(len(hot), generator=gen)[: n // 2]], torch.randperm(rows, generator=gen)[: n // 2]]).unique()
                    ids = torch.cat([mix[mix % G == o] for o in range(G)]).cuda()                      # bucketed by owner
                    cnt = torch.tensor([int((mix % G == o).sum()) for o in range(G)], dtype=torch.int64).cuda()
                    out = torch.empty((ids.numel(), dim), dtype=torch.float32, device="cuda")
                    stream.synchronize()
                    bar.wait()
                    exs[r].fetch_bucketed(caches[r], out.data_ptr(), ids.data_ptr(), ids.numel(), cnt.data_ptr())
                    stream.synchronize()
                    if step == args.steps - 1:
                        assert torch.equal(out, feature_rows_torch(ids, dim, 0))
                    bar.wait()
        except BaseException as e:  # noqa: BLE001
            errors.append((r, repr(e)))
            bar.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errors:
        raise SystemExit(str(errors))
    hm = [c.stats()[:2] for c in caches]
    print(f"# workload: {G} logical ranks, {n} ids each per step, {args.steps} steps, {args.rounds} exchange rounds; owner hit/miss totals {hm}; last step checked bit-exact", flush=True)
    for e in exs:
        e.close()
    _capi.check(L.coala_comm_group_destroy(group))
    for c in caches:
        c.close()
    for t in shards:
        t.close()


def summarize(path, last_steps):
    rows = list(csv.DictReader(open(path)))
    ev = []
    for r in rows:
        name = r["Kernel_Name"]
        kind = "fill" if "miss_fill_kernel" in name else ("copy" if ("copyBuffer" in name or "inproc_copy_kernel" in name) else ("probe" if "probe_gather_kernel" in name else None))
        if kind:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Queue_Id", "?")))
    ev.sort()
    fills = [e for e in ev if e[2] == "fill" and e[1] - e[0] > 20000]                      # cold fills that move rows (> 20 us)
    copies = [e for e in ev if e[2] == "copy" and e[1] - e[0] > 3000]                      # row copies (the id copies are shorter)
    fills = fills[-last_steps:]
    print(f"# {len(fills)} cold-fill launches (last of the run), {len(copies)} row copies in the whole trace; times in us relative to each fill's start")
    for f0, f1, _, q in fills:
        started = [(c0 - f0, c1 - f0) for c0, c1, _, cq in copies if f0 <= c0 < f1]
        show = ", ".join(f"[{a / 1e3:.0f}..{b / 1e3:.0f}]" for a, b in started[:6])
        print(f"fill on queue {q}: {(f1 - f0) / 1e3:7.1f} us; row copies that STARTED while it ran: {len(started):2d}  {show}")
    first, last = fills[0][0], fills[-1][1]
    # every row copy of the window against the UNION of the fill intervals (two ranks' fills can run at once)
    iv = sorted((f0, f1) for f0, f1, _, _ in fills)
    merged = []
    for a0, a1 in iv:
        if merged and a0 <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], a1)
        else:
            merged.append([a0, a1])
    win = [(c0, c1) for c0, c1, _, _ in copies if first <= c0 <= last]
    cp_ns = sum(c1 - c0 for c0, c1 in win)
    ov_ns = sum(max(0, min(c1, m1) - max(c0, m0)) for c0, c1 in win for m0, m1 in merged)
    began_inside = sum(1 for c0, c1 in win if any(m0 <= c0 < m1 for m0, m1 in merged))
    print(f"# in that window: {len(win)} row copies, {began_inside} of them began while a cold fill was running; "
          f"{ov_ns / 1e3:.0f} us of their {cp_ns / 1e3:.0f} us ran beside a fill ({100.0 * ov_ns / max(cp_ns, 1):.0f} %)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--summarize", type=str, default=None)
    ap.add_argument("--last", type=int, default=16)
    a = ap.parse_args()
    if a.summarize:
        summarize(a.summarize, a.last)
    else:
        run(a)
