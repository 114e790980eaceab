#!/usr/bin/env python3
"""Timing of the owner-partitioned path's device pieces on ONE GPU (development tool): route, serve, scatter."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
import torch
import COALA_GNN_Pybind as P
from COALA_GNN.synthetic import alloc_pinned_table

torch.cuda.set_device(0)
G, dim, rows, n = 8, 1024, 2_000_000, 28500
table = alloc_pinned_table(rows, dim, 0, 0)
ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
cache = P.SSD_GNN_NVSHMEM_Cache(ctrl, None, 0, G, 4096, table.device_ptr, num_rows=rows, sync=False)
idx = torch.randperm(rows, device="cuda")[:n]
node = torch.empty(n, dtype=torch.int64, device="cuda"); mp = torch.empty_like(node)
cnt = torch.zeros(G, dtype=torch.int64, device="cuda"); off = torch.zeros(G + 1, dtype=torch.int64, device="cuda")
out = torch.empty((n, dim), dtype=torch.float32, device="cuda"); src = torch.rand((n, dim), device="cuda")
own = idx[idx % G == 0]
def timeit(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("route   us", round(timeit(lambda: cache.route(idx.data_ptr(), n, G, node.data_ptr(), mp.data_ptr(), cnt.data_ptr(), off.data_ptr(), 0)), 2))
print("scatter us", round(timeit(lambda: cache.scatter(out.data_ptr(), src.data_ptr(), mp.data_ptr(), n)), 2), "->", round(2 * n * dim * 4 / 1e3 / timeit(lambda: cache.scatter(out.data_ptr(), src.data_ptr(), mp.data_ptr(), n)), 1), "GB/s")
ids8 = torch.randperm(rows // G, device="cuda")[:n] * G
cache.serve(out.data_ptr(), ids8.data_ptr(), n); torch.cuda.synchronize()
print("serve (all hits) us", round(timeit(lambda: cache.serve(out.data_ptr(), ids8.data_ptr(), n)), 2))
