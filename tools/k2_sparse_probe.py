#!/usr/bin/env python3
"""Cold fill (K2) over PCIe when only a few rows miss -- the 8-GPU steady state (aggregate cache = 84 % of the table): achieved PCIe
rate of miss_fill_kernel against the miss ratio of a 28,500-row batch (pinned-host cold tier, 4-KiB rows).  Development tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
if os.environ.get("COALA_K2_UNIT_TILES") is not None and os.environ.get("K2_SPARSE") is None:
    os.environ["K2_SPARSE"] = "32"             # (the product's threshold; COALA_K2_UNIT_TILES=0 is the static deal of rounds 1-3, a knob of the development library)
if os.environ.get("K2_SPARSE") is not None:   # development library: the compaction threshold is a knob there
    os.environ["COALA_HIP_LIB"] = os.path.join(ROOT, "coala-gnn_amd", "lib", "libcoala_hip_dev.so")
    os.environ["COALA_K2_SPARSE"] = os.environ["K2_SPARSE"]
import torch
import COALA_GNN_Pybind as P
from COALA_GNN.synthetic import alloc_pinned_table, feature_rows_torch
torch.cuda.set_device(0)
rows, dim, n = 2_000_000, 1024, 28500
table = alloc_pinned_table(rows, dim, 0, 0)
ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
perm = torch.randperm(rows, device="cuda")
out = torch.empty((n, dim), dtype=torch.float32, device="cuda")
print(f"# tools/k2_sparse_probe.py: 28,500-row batch, pinned-host cold tier; K2 by events attached to the launch; compaction threshold {os.environ.get('K2_SPARSE', 'product default')}, verdict tile rows {os.environ.get('COALA_K2_TILE_ROWS', 'product default (64)')}")
for miss_pct in (100, 68, 32, 16, 8, 4, 2):
    res = []
    for rep in range(5):
        cache = P.Isolated_Cache(ctrl, None, 0, 1, 4096, table.device_ptr, num_rows=rows, profile=True, sync=False, max_batch=n)
        warm = perm[rep * 4 * n: rep * 4 * n + n].contiguous()
        cold = perm[rep * 4 * n + n: rep * 4 * n + 2 * n].contiguous()
        cache.read_feature(out.data_ptr(), warm.data_ptr(), n)
        k = n * miss_pct // 100
        ids = torch.cat([cold[:k], warm[: n - k]])[torch.randperm(n, device="cuda")].contiguous()
        torch.cuda.synchronize()
        cache.profile(reset=True)
        cache.read_feature(out.data_ptr(), ids.data_ptr(), n)
        torch.cuda.synchronize()
        p = cache.profile()
        if rep == 0:   # the rows this launch shape delivered == the table's formula (the knobs of the development library change who streams what, never the result)
            assert torch.equal(out, feature_rows_torch(ids, dim, 0)), f"rows differ from the table at {miss_pct} % misses"
        res.append(p.fill_ms / max(p.fill_launches, 1) * 1e3)
        cache.close()
    us = sorted(res)[len(res) // 2]
    mb = k * dim * 4 / 1e6
    print(f"miss {miss_pct:3d} % ({k:6d} rows, {mb:7.2f} MB over PCIe): K2 {us:8.1f} us = {mb / us * 1e3:6.2f} GB/s", flush=True)
