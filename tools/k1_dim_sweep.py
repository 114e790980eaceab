#!/usr/bin/env python3
"""probe_gather_kernel hit-ratio sweep for every line size (cache_dim 128 / 256 / 512 / 1024): the product call
(coala_cache_read_feature, COALA_FLAG_PROFILE: events attached to the K1 launch) on a batch of unique ids of which a given share is cached.
Cold tier in HBM so that the fill between the timed launches is short; the batch is re-warmed before every measurement."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
import torch
import COALA_GNN_Pybind as P
torch.cuda.set_device(0)
print("# tools/k1_dim_sweep.py: K1 alone (HIP events attached to the launch: the kernel's own begin -> end), algorithmic bytes = rows x 264 B + hits x 2 x dim x 4 B")
for dim, n, rows in ((128, 1081344, 8_000_000), (256, 262144, 4_000_000), (512, 123904, 4_000_000), (1024, 36864, 2_000_000), (1024, 123904, 2_000_000)):
    table = torch.rand((rows, dim), dtype=torch.float32, device="cuda")
    ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
    perm = torch.randperm(rows, device="cuda")
    warm, cold = perm[:n].contiguous(), perm[n: 2 * n].contiguous()
    out = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    for hit in (0, 32, 75, 100):
        cache = P.Isolated_Cache(ctrl, None, 0, 1, 4096, table.data_ptr(), num_rows=rows, profile=True, sync=False, max_batch=n)
        cache.read_feature(out.data_ptr(), warm.data_ptr(), n)          # cache exactly the warm ids
        k = n * hit // 100
        ids = torch.cat([warm[:k], cold[: n - k]])[torch.randperm(n, device="cuda")].contiguous()
        torch.cuda.synchronize()
        us = []
        for rep in range(12):
            cache.profile(reset=True)
            cache.read_feature(out.data_ptr(), ids.data_ptr(), n)
            torch.cuda.synchronize()
            p = cache.profile()
            us.append(p.gather_ms / max(p.gather_launches, 1) * 1e3)
            if hit != 100:   # the launch just timed cached the cold ids: only the first one sees the stated hit ratio
                break
        t = sorted(us)[len(us) // 2]
        alg = n * 264 + k * 2 * dim * 4
        print(f"dim {dim:5d} n={n:8d} hit {hit:3d} %: K1 {t:8.2f} us   {alg / t / 1e3:7.1f} GB/s = {alg / t / 1e3 / 80:5.1f} % of 8 TB/s", flush=True)
        cache.close()
    del table
