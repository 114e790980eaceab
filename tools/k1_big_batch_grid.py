#!/usr/bin/env python3
"""K1 on the large all-hit batches (123,904 x 4 KiB, 262,144 x 1 KiB, 1,081,344 x 512 B) against the grid size: development build,
COALA_K1_GRID from the environment (one process per setting: tools/k1_big_batch_grid.sh).  Development tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
os.environ["COALA_HIP_LIB"] = os.path.join(ROOT, "coala-gnn_amd", "lib", "libcoala_hip_dev.so")
import torch
import COALA_GNN_Pybind as P
torch.cuda.set_device(0)
tag = "GRID=" + os.environ.get("COALA_K1_GRID", "default") + " WAVES=" + os.environ.get("COALA_K1_WAVES", "default")
for dim, n, rows in ((1024, 123904, 2_000_000), (256, 262144, 4_000_000), (128, 1081344, 8_000_000), (1024, 36864, 2_000_000)):
    table = torch.rand((rows, dim), dtype=torch.float32, device="cuda")
    ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
    perm = torch.randperm(rows, device="cuda")
    warm = perm[:n].contiguous()
    outs = [torch.empty((n, dim), dtype=torch.float32, device="cuda") for _ in range(3)]   # in rotation: a reused buffer is partly still in the Infinity Cache when it is overwritten
    out = outs[0]
    for hit in (90, 100):
        cache = P.Isolated_Cache(ctrl, None, 0, 1, 4096, table.data_ptr(), num_rows=rows, profile=True, sync=False, max_batch=n)
        cache.read_feature(out.data_ptr(), warm.data_ptr(), n)
        k = n * hit // 100
        us = []
        for rep in range(3 if hit < 100 else 8):
            cold = perm[n * (1 + rep): n * (2 + rep)]
            ids = torch.cat([warm[:k], cold[: n - k]])[torch.randperm(n, device="cuda")].contiguous()
            torch.cuda.synchronize()
            cache.profile(reset=True)
            cache.read_feature(outs[(rep + 1) % 3].data_ptr(), ids.data_ptr(), n)
            torch.cuda.synchronize()
            p = cache.profile()
            us.append(round(p.gather_ms / max(p.gather_launches, 1) * 1e3, 1))
        t = sorted(us)[len(us) // 2]
        alg = n * (8 + cache.geometry().tag_set_bytes) + k * 2 * dim * 4
        print(f"{tag:28s} dim {dim:5d} n={n:8d} hit {hit:3d} %: K1 {t:8.2f} us = {alg / t / 1e3 / 80:5.1f} % of 8 TB/s   all: {us}", flush=True)
        cache.close()
    del table, out, outs
