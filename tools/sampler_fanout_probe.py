"""Sampler wall time per call for the BASELINE fan-outs (development tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
import torch
from COALA_GNN.sampler import NeighborSampler
from COALA_GNN.synthetic import powerlaw_csc
rows = 10_000_000
torch.cuda.set_device(0)
indptr, indices = powerlaw_csc(rows, 12.0, seed=0, device="cuda")
ids = torch.randperm(6_000_000, device="cuda")
for fan, G in (([5, 5], 0), ([5, 5], 8), ([10, 10], 0), ([10, 10], 8), ([15, 10, 5], 0), ([10, 10, 10], 0), ([10, 10, 10], 8)):
    s = NeighborSampler(fan, bucket_by_owner=G); g = s.make_graph(indptr, indices)
    for it in range(10): s.sample(g, ids[it * 1024:(it + 1) * 1024])
    torch.cuda.synchronize(); t0 = time.perf_counter(); N = 100; n_in = 0
    for it in range(N):
        n_in += s.sample(g, ids[(it + 10) * 1024:(it + 11) * 1024])[0].numel()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / N * 1e3
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
    for it in range(N):
        ev[it][0].record(); s.sample(g, ids[(it + 10) * 1024:(it + 11) * 1024]); ev[it][1].record()
    torch.cuda.synchronize()
    gpu = sorted(a.elapsed_time(b) for a, b in ev)[N // 2]
    print(f"fanout {fan}{' bucketed by 8 owners' if G else ''}: {wall:.3f} ms per call (host wall), {gpu:.3f} ms on the stream (HIP events, median), {n_in / N:.0f} input nodes")
