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
TAG64 = os.environ.get("TAG64") == "1"
print(f"# tools/k1_dim_sweep.py: K1 alone (HIP events attached to the launch: the kernel's own begin -> end), algorithmic bytes = rows x (8 + {256 if TAG64 else 128}) B + hits x 2 x dim x 4 B  ({64 if TAG64 else 32}-bit tags)")
SHAPES = ((128, 1081344, 8_000_000), (256, 262144, 4_000_000), (512, 123904, 4_000_000), (1024, 36864, 2_000_000), (1024, 123904, 2_000_000))
if os.environ.get("SHAPES"):   # "dim:n:rows,dim:n:rows": other batch sizes / table sizes (CACHE_MB sets the cache size, default 4096)
    SHAPES = tuple(tuple(int(x) for x in sh.split(":")) for sh in os.environ["SHAPES"].split(","))
HITS = tuple(int(h) for h in os.environ.get("HITS", "0,25,32,50,75,90,100").split(","))
for dim, n, rows in SHAPES:
    table = torch.rand((rows, dim), dtype=torch.float32, device="cuda")
    ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
    perm = torch.randperm(rows, device="cuda")
    warm = perm[:n].contiguous()
    outs = [torch.empty((n, dim), dtype=torch.float32, device="cuda") for _ in range(3)]   # in rotation: a reused buffer is partly still in the Infinity Cache when it is overwritten
    out = outs[0]
    for hit in HITS:   # 0.25 / 0.5 / 0.75 / 0.9, plus the default workload (0.32) and the two ends
        cache = P.Isolated_Cache(ctrl, None, 0, 1, int(os.environ.get('CACHE_MB', 4096)), table.data_ptr(), num_rows=rows, profile=True, sync=False, max_batch=n, tag64=TAG64)
        cache_tag_bytes = cache.geometry().tag_set_bytes
        cache.read_feature(out.data_ptr(), warm.data_ptr(), n)          # cache exactly the warm ids
        k = n * hit // 100
        # below 100 % the launch just timed caches its cold ids: every repetition takes fresh ones (as long as the sets stay far from
        # full, so that no warm line is evicted: 3 repetitions, 1 for the 1 M-row batch of 512-B lines)
        reps = 12 if hit == 100 else (3 if (n * 4 <= cache.geometry().num_sets * 32 // 4 and 5 * n <= rows) else 1)
        us = []
        for rep in range(reps):
            cold = perm[n * (1 + rep): n * (2 + rep)]
            ids = torch.cat([warm[:k], cold[: n - k]])[torch.randperm(n, device="cuda")].contiguous()
            torch.cuda.synchronize()
            cache.profile(reset=True)
            cache.read_feature(outs[(rep + 1) % 3].data_ptr(), ids.data_ptr(), n)
            torch.cuda.synchronize()
            p = cache.profile()
            us.append(p.gather_ms / max(p.gather_launches, 1) * 1e3)
        t = sorted(us)[len(us) // 2]
        alg = n * (8 + cache_tag_bytes) + k * 2 * dim * 4
        print(f"dim {dim:5d} n={n:8d} hit {hit:3d} %: K1 {t:8.2f} us   {alg / t / 1e3:7.1f} GB/s = {alg / t / 1e3 / 80:5.1f} % of 8 TB/s", flush=True)
        cache.close()
    del table
