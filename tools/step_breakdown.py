#!/usr/bin/env python3
"""Wall-clock breakdown of one serial training step of the loader (development tool)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
import numpy as np, torch
from COALA_GNN import MPI_Comm_Manager, Node_Distributor, SSD_INFO, COALA_GNN_DataLoader
from COALA_GNN.harness import SageMean
from COALA_GNN.sampler import NeighborSampler
from COALA_GNN.synthetic import alloc_pinned_table, block_colors, powerlaw_csc

rows, dim, batch, fan = 4_000_000, 1024, 1024, [5, 5]
torch.cuda.set_device(0)
table = alloc_pinned_table(rows, dim, 0, 0)
indptr, indices = powerlaw_csc(rows, 12.0, seed=0, device="cuda")
comm = MPI_Comm_Manager(0); comm.initialize_nested_process_group("isolated")
tmp = tempfile.mkdtemp()
color, tk, sc, _ = block_colors(rows)
files = [os.path.join(tmp, f) for f in ("color.npy", "topk.npy", "score.npy")]
np.save(files[0], color); np.save(files[1], tk); np.save(files[2], sc)
ids = torch.randperm(int(0.6 * rows), generator=torch.Generator().manual_seed(0))[: 600 * batch]
nd = Node_Distributor(comm, ids, batch, *files, parsing_method="baseline")
sampler = NeighborSampler(fan, seed=0)
g = sampler.make_graph(indptr, indices, ndata={"labels": (torch.arange(rows, device="cuda") * 7) % 19})
loader = COALA_GNN_DataLoader(SSD_INFO(1, 4096, 1024, 0), nd, g, sampler, batch, dim, fan, 4096, "cuda:0", cache_backend="isolated", sim_buf=table, num_rows=rows)
model = SageMean(dim, 128, 19).cuda(); opt = torch.optim.Adam(model.parameters(), 1e-3); lossf = torch.nn.CrossEntropyLoss()
T = {k: 0.0 for k in ("sched", "to_dev", "sample", "fetch", "train")}
def sync(): torch.cuda.synchronize()
N = 300
for step in range(N + 100):
    if step == 100:
        T = {k: 0.0 for k in T}
    sync(); t0 = time.perf_counter()
    seeds_cpu = loader.scheduler.run(False); t1 = time.perf_counter()
    seeds = seeds_cpu.to("cuda:0"); sync(); t2 = time.perf_counter()
    b = sampler.sample(g, seeds); sync(); t3 = time.perf_counter()
    inp, sd, blocks, feat = loader.COALA_GNN_Manager.fetch_feature(b); sync(); t4 = time.perf_counter()
    loss = lossf(model(blocks, feat), blocks[-1].dstdata["labels"].view(-1)); opt.zero_grad(); loss.backward(); opt.step(); sync(); t5 = time.perf_counter()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): T[k] += v
print({k: round(v / N * 1e3, 3) for k, v in T.items()}, "ms per step; total", round(sum(T.values()) / N * 1e3, 3))
