#!/usr/bin/env python3
"""Device side of ONE rank's share of an 8-way distributed step on one GPU, every row a hit (development tool).

  --mode before : the round-1 sequence -- route, serve ALL received rows into the staging buffer, the all-to-all's self segment
                  (a device copy staging -> receive buffer), un-permute ALL rows
  --mode after  : the split-phase sequence -- route, probe with the own segment redirected into the output tensor, fills in
                  rounds, un-permute of the REMOTE rows only (own-shard rows never touch a staging buffer)
  --mode bucketed : ids delivered bucketed by owner by the sampler (coala_cache_fetch_distributed_bucketed) -- no route, the own
                  bucket gathered in place, remote rows received in place: nothing to un-permute
Prints the time of each piece (HIP events) and the algorithmic HBM bytes of the step; under
`rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/dist_breakdown.py --mode X --reps 20 --pmc`
the per-kernel counters give the measured bytes (tools/pmc_by_kernel.py sums them)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
import torch  # noqa: E402
import COALA_GNN_Pybind as P  # noqa: E402
from COALA_GNN.synthetic import alloc_pinned_table  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="both", choices=["before", "after", "bucketed", "both"])
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--pmc", action="store_true", help="no per-piece timing loops: run the whole step --reps times (for a counter pass)")
ap.add_argument("--rounds", type=int, default=2)
args = ap.parse_args()

torch.cuda.set_device(0)
G, me, dim, rows, n = 8, 0, 1024, 2_000_000, 28500
table = alloc_pinned_table(rows, dim, 0, 0)
ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
cache = P.SSD_GNN_NVSHMEM_Cache(ctrl, None, me, G, 4096, table.device_ptr, num_rows=rows, sync=False, rank=me)
gen = torch.Generator(device="cuda").manual_seed(1)
idx = torch.randperm(rows, device="cuda", generator=gen)[:n]                      # this rank's minibatch
node = torch.empty(n, dtype=torch.int64, device="cuda")
mp = torch.empty_like(node)
cnt = torch.zeros(G, dtype=torch.int64, device="cuda")
off = torch.zeros(G + 1, dtype=torch.int64, device="cuda")
out = torch.empty((n, dim), dtype=torch.float32, device="cuda")
cache.route(idx.data_ptr(), n, G, node.data_ptr(), mp.data_ptr(), cnt.data_ptr(), off.data_ptr(), 0)
scnt, sdis = cnt.cpu().tolist(), off.cpu().tolist()
# what this owner receives: its own bucket in place `me`, and from every other rank a bucket of the same size (ids it owns)
own_ids = node[sdis[me]: sdis[me] + scnt[me]]
others = (torch.randperm(rows // G, device="cuda", generator=gen)[: n - scnt[me]] * G + me)
per = [(n - scnt[me]) // (G - 1)] * (G - 1)
per[-1] += (n - scnt[me]) - sum(per)
rcnt = per[:me] + [scnt[me]] + per[me:]
rdis = [sum(rcnt[:p]) for p in range(G)]
pieces, pos = [], 0
for p in range(G):
    if p == me:
        pieces.append(own_ids)
    else:
        pieces.append(others[pos: pos + rcnt[p]])
        pos += rcnt[p]
recv_ids = torch.cat(pieces).contiguous()
tot = recv_ids.numel()
rows_send = torch.empty((tot, dim), dtype=torch.float32, device="cuda")
rows_recv = torch.rand((n, dim), device="cuda")
cache.serve(rows_send.data_ptr(), recv_ids.data_ptr(), tot)                      # warm: every id below is a hit from now on
torch.cuda.synchronize()
assert cache.stats(reset=True)[1] > 0

K = args.rounds
fill = [[(rdis[p] + rcnt[p] * k // K, rdis[p] + rcnt[p] * (k + 1) // K) for p in range(G) if p != me] for k in range(K)]
fill[-1].append((rdis[me], rdis[me] + rcnt[me]))
land = [[(sdis[p] + scnt[p] * k // K, sdis[p] + scnt[p] * (k + 1) // K) for p in range(G) if p != me] for k in range(K)]


def route():
    cache.route(idx.data_ptr(), n, G, node.data_ptr(), mp.data_ptr(), cnt.data_ptr(), off.data_ptr(), 0)


def serve_before():
    cache.serve(rows_send.data_ptr(), recv_ids.data_ptr(), tot)


def self_copy():
    rows_recv[sdis[me]: sdis[me] + scnt[me]].copy_(rows_send[rdis[me]: rdis[me] + rcnt[me]])


def scatter_before():
    cache.scatter(out.data_ptr(), rows_recv.data_ptr(), mp.data_ptr(), n)


def serve_after():
    cache.serve_probe_redirect(rows_send.data_ptr(), recv_ids.data_ptr(), tot, rdis[me], rdis[me] + rcnt[me], out.data_ptr(),
                               mp.data_ptr() + sdis[me] * 8)
    for k in range(K):
        cache.serve_fill_ranges(rows_send.data_ptr(), recv_ids.data_ptr(), tot, fill[k])


def serve_bucketed():   # the own bucket sits at its offset of `out`, in order (row_map = NULL)
    cache.serve_probe_redirect(rows_send.data_ptr(), recv_ids.data_ptr(), tot, rdis[me], rdis[me] + rcnt[me], out.data_ptr() + sdis[me] * dim * 4, 0)
    for k in range(K):
        cache.serve_fill_ranges(rows_send.data_ptr(), recv_ids.data_ptr(), tot, fill[k])


def scatter_after():
    for k in range(K):
        cache.scatter_ranges(out.data_ptr(), rows_recv.data_ptr(), mp.data_ptr(), land[k])


def timeit(fn, reps):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


row = dim * 4
probe_b = tot * (8 + 128)   # 32-bit tags: one 128-B line per set
modes = {"before": [("route", route, 0), ("serve -> staging", serve_before, probe_b + 2 * tot * row),
                    ("self segment copy", self_copy, 2 * scnt[me] * row), ("un-permute all rows", scatter_before, 2 * n * row + 8 * n)],
         "after": [("route", route, 0), ("probe+fills (own shard -> out)", serve_after, probe_b + 2 * tot * row + 8 * scnt[me]),
                   ("un-permute remote rows", scatter_after, 2 * (n - scnt[me]) * row + 8 * (n - scnt[me]))],
         "bucketed": [("probe+fills (own bucket in place)", serve_bucketed, probe_b + 2 * tot * row)]}
print(f"G={G} rank {me}: batch {n} rows x {row} B, own bucket {scnt[me]} rows, owner batch {tot} rows (all hits), rounds {K}")
for mode in (["before", "after", "bucketed"] if args.mode == "both" else [args.mode]):
    if args.pmc:
        for _ in range(args.reps):
            for _, fn, _ in modes[mode]:
                fn()
        torch.cuda.synchronize()
        print(f"{mode}: {args.reps} steps issued")
        continue
    total_us, total_b = 0.0, 0
    for name, fn, nbytes in modes[mode]:
        us = timeit(fn, args.reps)
        total_us += us
        total_b += nbytes
        print(f"  {mode:8s} {name:34s} {us:8.2f} us   {nbytes / 1e6:8.2f} MB algorithmic HBM traffic" + (f"   {nbytes / us / 1e3:7.1f} GB/s" if nbytes else ""))
    print(f"  {mode:8s} {'TOTAL (device kernels, one rank)':34s} {total_us:8.2f} us   {total_b / 1e6:8.2f} MB per step")
hit, miss, _ = cache.stats()
assert miss == 0, (hit, miss)
