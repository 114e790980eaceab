"""Does the sampler (stream B) overlap a running cold-tier fill (stream A)?  Development tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
import torch
from COALA_GNN_Pybind import Isolated_Cache, SSD_GNN_SSD_Controllers
from COALA_GNN.sampler import NeighborSampler
from COALA_GNN.synthetic import alloc_pinned_table, powerlaw_csc
rows, dim = 2_000_000, 1024
torch.cuda.set_device(0)
table = alloc_pinned_table(rows, dim, 0, 0)
indptr, indices = powerlaw_csc(rows, 12.0, seed=0, device="cuda")
s = NeighborSampler([5, 5]); g = s.make_graph(indptr, indices)
ctrl = SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
cache = Isolated_Cache(ctrl, None, 0, 1, 256, table.data_ptr(), num_rows=rows, sync=False, max_batch=40000)
A, B = torch.cuda.Stream(), torch.cuda.Stream()
out = torch.empty(36000, dim, device="cuda")
clk = time.perf_counter
gen = torch.Generator().manual_seed(0)
pin = torch.zeros(16, dtype=torch.int64).pin_memory()
dv = torch.zeros(16, dtype=torch.int64, device="cuda")
def extra(kind):
    tot = 0.0; N = 100
    for it in range(N):
        idx = torch.randint(0, rows, (36000,), generator=gen).cuda()
        seeds = torch.randint(0, rows, (1024,), generator=gen)
        seeds_d = seeds.cuda()
        torch.cuda.synchronize()
        with torch.cuda.stream(A):
            cache.read_feature(out.data_ptr(), idx.data_ptr(), 36000)
        t0 = clk()
        with torch.cuda.stream(B):
            if kind == "tiny kernel + sync": dv.add_(1); B.synchronize()
            elif kind == "pinned H2D + sync": dv.copy_(pin, non_blocking=True); B.synchronize()
            elif kind == "pinned D2H + sync": pin.copy_(dv, non_blocking=True); B.synchronize()
            elif kind == "pageable H2D": seeds.cuda()
            elif kind == "sampler, device seeds": s.sample(g, seeds_d)
        tot += clk() - t0
        torch.cuda.synchronize()
    print(f"while fill runs, {kind}: {tot / N * 1e6:.1f} us")
for k in ("tiny kernel + sync", "pinned H2D + sync", "pinned D2H + sync", "pageable H2D", "sampler, device seeds"):
    extra(k)
for mode in ("sampler alone", "sampler while fill runs", "fill alone"):
    tot = 0.0; N = 100
    for it in range(N):
        idx = torch.randint(0, rows, (36000,), generator=gen).cuda()
        seeds = torch.randint(0, rows, (1024,), generator=gen)
        torch.cuda.synchronize()
        t0 = clk()
        if mode != "sampler alone":
            with torch.cuda.stream(A):
                cache.read_feature(out.data_ptr(), idx.data_ptr(), 36000)
        if mode != "fill alone":
            with torch.cuda.stream(B):
                s.sample(g, seeds)
        t1 = clk()
        torch.cuda.synchronize()
        tot += (t1 - t0) if mode != "fill alone" else (clk() - t0)
    print(f"{mode}: {tot / N * 1e6:.1f} us host per call")
