#!/usr/bin/env python3
"""Steady-state launches of the two kernels whose wave-scheduler counters nobody had read: the un-permute kernel of a
distributed fetch (scatter_rows_kernel: out[map[r]] = src[r], 28.5 k rows x 4 KiB per launch, three destinations in rotation) and the
WIDE-grid form of the cold fill (miss_fill_kernel with the cold tier in HBM: 2048 blocks instead of the 20 of the host tier), on the
default workload's minibatches.  The harness for tools/k1_sq_counters.sh (HARNESS=tools/kernel_sq_probe.py); prints its own event timings.

  ROWS=10000000 DIM=1024 CACHE_MB=4096 python tools/kernel_sq_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
import torch  # noqa: E402
import COALA_GNN_Pybind as P  # noqa: E402
from COALA_GNN.sampler import NeighborSampler  # noqa: E402
from COALA_GNN.synthetic import feature_rows_torch, powerlaw_csc  # noqa: E402

rows, dim, batch, cache_mb = int(os.environ.get("ROWS", 10_000_000)), int(os.environ.get("DIM", 1024)), 1024, int(os.environ.get("CACHE_MB", 4096))
fanout = [5, 5]
torch.cuda.set_device(0)
dev = "cuda:0"
table = torch.empty((rows, dim), dtype=torch.float32, device=dev)         # the cold tier in HBM: K2 takes its wide grid
for lo in range(0, rows, 1 << 20):
    hi = min(rows, lo + (1 << 20))
    feature_rows_torch(torch.arange(lo, hi, device=dev), dim, 0, out=table[lo:hi])
indptr, indices = powerlaw_csc(rows, 12.0, seed=0, device=dev)
train_ids = torch.randperm(int(0.6 * rows), generator=torch.Generator().manual_seed(0))
sampler = NeighborSampler(fanout, seed=0)
graph = sampler.make_graph(indptr, indices)
batches = [sampler.sample(graph, train_ids[s * batch: (s + 1) * batch].cuda(), step=s)[0] for s in range(620)]
max_rows = batch * 36
ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
cache = P.Isolated_Cache(ctrl, None, 0, 1, cache_mb, table.data_ptr(), num_rows=rows, profile=True, sync=False, max_batch=max_rows)
outs = [torch.empty((max_rows, dim), dtype=torch.float32, device=dev) for _ in range(3)]
for b in batches[:420]:
    cache.read_feature(outs[0].data_ptr(), b.data_ptr(), b.numel())
torch.cuda.synchronize()
cache.stats(reset=True)
cache.profile(reset=True)
for k, b in enumerate(batches[420:]):
    cache.read_feature(outs[k % 3].data_ptr(), b.data_ptr(), b.numel())
torch.cuda.synchronize()
p = cache.profile()
hit, miss, _ = cache.stats()
assert torch.equal(outs[(len(batches) - 421) % 3][: batches[-1].numel()], table[batches[-1]])
fill_bytes = p.fill_rows / p.fill_launches * dim * 4 * 3       # cold row read + output row + line written (winners: nearly every miss)
print(f"cold tier in HBM: K1 {p.gather_ms / p.gather_launches * 1e3:7.2f} us   K2 (wide grid) {p.fill_ms / p.fill_launches * 1e3:7.2f} us per {p.fill_rows / p.fill_launches:.0f} missed rows = "
      f"{fill_bytes / (p.fill_ms / p.fill_launches * 1e-3) / 1e9:7.1f} GB/s of 3 x dim x 4 B per miss   hit {hit / (hit + miss):.4f}", flush=True)
# ---- the un-permute kernel alone: 28.5 k rows of a staging buffer to a random permutation of the rows of the output
n = 28544
src = torch.randn((n, dim), dtype=torch.float32, device=dev)
maps = [torch.randperm(n, device=dev) for _ in range(3)]
for k in range(20):
    cache.scatter(outs[k % 3].data_ptr(), src.data_ptr(), maps[k % 3].data_ptr(), n)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for k in range(200):
    cache.scatter(outs[k % 3].data_ptr(), src.data_ptr(), maps[k % 3].data_ptr(), n)
e.record()
e.synchronize()
us = a.elapsed_time(e) * 1e3 / 200
assert torch.equal(outs[199 % 3][maps[199 % 3]], src)
print(f"scatter_rows_kernel: {n} rows x {dim * 4} B: {us:7.2f} us per launch back to back (launch gap included) = {2 * n * dim * 4 / us / 1e3:7.1f} GB/s read + written", flush=True)
