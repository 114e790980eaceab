#!/usr/bin/env python3
"""Where the prefetching epoch loses time against the fetch-only floor: idle time of the FETCH STREAM between consecutive fetches
(HIP events: end of fetch t -> start of fetch t+1) under a training load, steady state (warm cache).  Development tool."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
import numpy as np, torch
from COALA_GNN import MPI_Comm_Manager, Node_Distributor, SSD_INFO, COALA_GNN_DataLoader
from COALA_GNN.harness import SageMean
from COALA_GNN.sampler import NeighborSampler
from COALA_GNN.synthetic import alloc_pinned_table, block_colors, powerlaw_csc
rows, dim, batch, fan = 10_000_000, 1024, 1024, [5, 5]
torch.cuda.set_device(0)
table = alloc_pinned_table(rows, dim, 0, 0)
indptr, indices = powerlaw_csc(rows, 12.0, seed=0, device="cuda")
comm = MPI_Comm_Manager(0); comm.initialize_nested_process_group("isolated")
tmp = tempfile.mkdtemp()
color, tk, sc, _ = block_colors(rows)
files = [os.path.join(tmp, f) for f in ("color.npy", "topk.npy", "score.npy")]
np.save(files[0], color); np.save(files[1], tk); np.save(files[2], sc)
steps = int(os.environ.get("STEPS", "2400"))
ids = torch.randperm(int(0.6 * rows), generator=torch.Generator().manual_seed(0))[: (steps + 1) * batch]
sampler = NeighborSampler(fan, seed=0)
g = sampler.make_graph(indptr, indices, ndata={"labels": (torch.arange(rows, device="cuda") * 7) % 19})
nd = Node_Distributor(comm, ids, batch, *files, parsing_method="baseline")
loader = COALA_GNN_DataLoader(SSD_INFO(1, 4096, 1024, 0), nd, g, sampler, batch, dim, fan, 4096, "cuda:0", cache_backend="isolated", sim_buf=table,
                              num_rows=rows, prefetch=int(os.environ.get("PREFETCH", "2")), refresh_counter=int(os.environ.get("REFRESH", "10")))
mgr = loader.COALA_GNN_Manager
pairs = []
def keep(wait):  # keep every (start, end) event pair instead of folding them away
    pairs.extend(mgr._agg_events); mgr._agg_events = []
mgr._fold_events = keep
model = SageMean(dim, 128, 19).cuda(); opt = torch.optim.Adam(model.parameters(), 1e-3, fused=True); lossf = torch.nn.CrossEntropyLoss()
train = os.environ.get("TRAIN", "1") == "1"
import gc; gc.collect(); gc.freeze()   # a full collection of the interpreter inside the loop is a 100 ms hole of its own
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
for inp, sd, blocks, feat in loader:
    if train:
        loss = lossf(model(blocks, feat), blocks[-1].dstdata["labels"].view(-1)); opt.zero_grad(); loss.backward(); opt.step()
    n += 1
torch.cuda.synchronize(); dt = time.perf_counter() - t0
keep(True)
warm = 1200  # steady state only
dur = [a.elapsed_time(b) for a, b in pairs[warm:]]
gap = [pairs[i][1].elapsed_time(pairs[i + 1][0]) for i in range(warm, len(pairs) - 1)]
import statistics as st
print(f"train={train} prefetch={loader.prefetch} refresh_counter={loader.refresh_counter}: {n} steps, {dt / n * 1e3:.3f} ms/step overall (cold start included)")
print(f"  steady state ({len(dur)} fetches): fetch duration mean {st.mean(dur):.4f} ms, median {st.median(dur):.4f} ms; idle gap between fetches mean {st.mean(gap):.4f} ms, "
      f"median {st.median(gap):.4f} ms; sum = {st.mean(dur) + st.mean(gap):.4f} ms/step")
big = sorted(gap, reverse=True)
for thr in (0.05, 0.2, 0.5, 1.0):
    sel = [x for x in gap if x > thr]
    print(f"  gaps > {thr:4.2f} ms: {len(sel):5d} ({100 * len(sel) / len(gap):5.1f} %), contributing {sum(sel) / len(gap):.4f} ms/step")
idx = [i for i, x in enumerate(gap) if x > 0.2][:40]
print("  positions (step mod 10) of the first gaps > 0.2 ms:", [(warm + i + 1) % 10 for i in idx])
