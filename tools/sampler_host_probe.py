"""Host-time breakdown of NeighborSampler.sample (development tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
import ctypes as C
import torch
from COALA_GNN.sampler import NeighborSampler, Block, _lib, _capi, current_stream
from COALA_GNN.synthetic import powerlaw_csc
rows = 10_000_000
torch.cuda.set_device(0)
indptr, indices = powerlaw_csc(rows, 12.0, seed=0, device="cuda")
s = NeighborSampler([5, 5]); g = s.make_graph(indptr, indices, ndata={"labels": (torch.arange(rows, device="cuda") * 7) % 19})
ids = torch.randperm(6_000_000)
clk = time.perf_counter
for stream in (None, torch.cuda.Stream()):
    T = dict(to=0.0, alloc=0.0, call=0.0, blocks=0.0, whole=0.0)
    N = 500
    with torch.cuda.stream(stream):
        for it in range(N + 50):
            if it == 50: T = {k: 0.0 for k in T}
            seeds_cpu = ids[it * 1024:(it + 1) * 1024].clone()
            t0 = clk()
            seeds = seeds_cpu.to("cuda:0"); t1 = clk()
            n = 1024; rev = [5, 5]; L = 2; caps = [n, n * 6, n * 36]
            src = [torch.empty(caps[l + 1], dtype=torch.int64, device="cuda") for l in range(L)]
            nbr = [torch.empty(caps[l] * 5, dtype=torch.int32, device="cuda") for l in range(L)]
            src_p = (C.c_void_p * L)(*[t.data_ptr() for t in src]); nbr_p = (C.c_void_p * L)(*[t.data_ptr() for t in nbr])
            fan = (C.c_int32 * L)(*rev); n_src = (C.c_int64 * L)(); t2 = clk()
            _capi.check(_lib.coala_sampler_sample(g._h, seeds.data_ptr(), n, fan, L, 0, it, src_p, nbr_p, n_src, current_stream())); t3 = clk()
            blocks = []; n_dst = n
            for l in range(L):
                ns = int(n_src[l])
                blocks.insert(0, Block(src[l][:ns], nbr[l][: n_dst * rev[l]].view(n_dst, rev[l]), n_dst, graph=g if l == 0 else None))
                n_dst = ns
            t4 = clk()
            s.sample(g, seeds_cpu); t5 = clk()
            for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): T[k] += v
    print("stream", "default" if stream is None else "side", {k: round(v / N * 1e6, 1) for k, v in T.items()}, "us")
