"""The prefetching loader with and without a training load: host-side split of a step (consumer wait, producer stages).
Development tool.  K2_GRID=<blocks> selects the development library and its cold-fill grid knob; ONLY_TRAIN=1 skips the no-training leg."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
if os.environ.get("K2_GRID"):
    os.environ["COALA_HIP_LIB"] = os.path.join(ROOT, "coala-gnn_amd", "lib", "libcoala_hip_dev.so")
    os.environ["COALA_K2_GRID"] = os.environ["K2_GRID"]
import numpy as np, torch
from COALA_GNN import MPI_Comm_Manager, Node_Distributor, SSD_INFO, COALA_GNN_DataLoader
from COALA_GNN.harness import SageMean
from COALA_GNN.sampler import NeighborSampler
from COALA_GNN.synthetic import alloc_pinned_table, block_colors, powerlaw_csc
import sys as _s
_s.setswitchinterval(float(os.environ.get("SWITCH", "0.005")))
rows, dim, batch, fan = 10_000_000, 1024, 1024, [5, 5]
torch.cuda.set_device(0)
table = alloc_pinned_table(rows, dim, 0, 0)
indptr, indices = powerlaw_csc(rows, 12.0, seed=0, device="cuda")
comm = MPI_Comm_Manager(0); comm.initialize_nested_process_group("isolated")
tmp = tempfile.mkdtemp()
color, tk, sc, _ = block_colors(rows)
files = [os.path.join(tmp, f) for f in ("color.npy", "topk.npy", "score.npy")]
np.save(files[0], color); np.save(files[1], tk); np.save(files[2], sc)
ids = torch.randperm(int(0.6 * rows), generator=torch.Generator().manual_seed(0))[: 1200 * batch]
sampler = NeighborSampler(fan, seed=0)
g = sampler.make_graph(indptr, indices, ndata={"labels": (torch.arange(rows, device="cuda") * 7) % 19})
for train in ((True,) if os.environ.get("ONLY_TRAIN") else (False, True)):
    nd = Node_Distributor(comm, ids, batch, *files, parsing_method="baseline")
    loader = COALA_GNN_DataLoader(SSD_INFO(1, 4096, 1024, 0), nd, g, sampler, batch, dim, fan, 4096, "cuda:0", cache_backend="isolated", sim_buf=table, num_rows=rows, prefetch=int(os.environ.get("PREFETCH", "2")))
    model = SageMean(dim, 128, 19).cuda(); opt = torch.optim.Adam(model.parameters(), 1e-3, fused=bool(int(os.environ.get("FUSED_ADAM", "1")))); lossf = torch.nn.CrossEntropyLoss()
    n = 0; wait = 0.0; t_train = 0.0
    ST = {}
    def wrap(obj, name, key):
        f = getattr(obj, name)
        def g(*a, **k):
            t = time.perf_counter()
            try: return f(*a, **k)
            finally: ST[key] = ST.get(key, 0.0) + time.perf_counter() - t
        setattr(obj, name, g)
    wrap(loader.COALA_GNN_Manager.COALA_GNN_Cache, "get_cache_data", "get_cache_data")
    wrap(nd, "parse_domain_training_nodes", "parse(in pool)")
    wrap(nd, "gather_cache_meta", "gather_meta(in pool)")
    wrap(comm, "broadcast_training_nodes", "broadcast")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    it = iter(loader)
    while True:
        a = time.perf_counter()
        try: inp, sd, blocks, feat = next(it)
        except StopIteration: break
        b = time.perf_counter(); wait += b - a
        if train:
            loss = lossf(model(blocks, feat), blocks[-1].dstdata["labels"].view(-1)); opt.zero_grad(); loss.backward(); opt.step()
        t_train += time.perf_counter() - b
        n += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"train={train}: {n} steps, {dt/n*1e3:.3f} ms/step, consumer wait {wait/n*1e3:.3f} ms/step, consumer python in train {t_train/n*1e3:.3f} ms/step")
    print("   producer host ms/step:", {k: round(v / n * 1e3, 3) for k, v in loader.producer_times.items()})
    print("   scheduler pieces ms/step:", {k: round(v / n * 1e3, 3) for k, v in ST.items()})
    del loader, nd
