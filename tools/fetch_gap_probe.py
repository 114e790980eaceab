#!/usr/bin/env python3
"""Where an epoch loses time against the fetch-only floor: idle time of the FETCH STREAM between consecutive fetches (HIP events: end of
fetch t -> start of fetch t+1) under a training load, steady state (warm cache) -- and, with WHY=1, what the host and the other streams
were doing during every large gap: the host-side calls of the loader (scheduler, sample_begin / sample_end, fetch enqueue) with their
wall-clock intervals, every garbage collection of the interpreter, every growth of the caching allocator's reserved memory, and the
completion time of the sample the late fetch waited for, all mapped onto one time axis.  Development tool.

  STEPS=2400 PREFETCH=0|2 REFRESH=10 TRAIN=1 WHY=1 GCFREEZE=1 TIMING_STRIDE=1|16 SAMPLER_WAIT=0|1 RECORD_STREAM=0|1 python tools/fetch_gap_probe.py
"""
import gc
import os
import statistics as st
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
import numpy as np, torch
from COALA_GNN import MPI_Comm_Manager, Node_Distributor, SSD_INFO, COALA_GNN_DataLoader
from COALA_GNN.harness import SageMean
from COALA_GNN.sampler import NeighborSampler
from COALA_GNN.synthetic import alloc_pinned_table, block_colors, powerlaw_csc

rows, dim, batch, fan = 10_000_000, 1024, 1024, [5, 5]
WHY = os.environ.get("WHY", "0") == "1"
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
kw = {}
loader = COALA_GNN_DataLoader(SSD_INFO(1, 4096, 1024, 0), nd, g, sampler, batch, dim, fan, 4096, "cuda:0", cache_backend="isolated", sim_buf=table,
                              num_rows=rows, prefetch=int(os.environ.get("PREFETCH", "2")), refresh_counter=int(os.environ.get("REFRESH", "10")), **kw)
mgr = loader.COALA_GNN_Manager
if os.environ.get("TIMING_STRIDE"):                 # 1: a timing pair on every fetch (needed for the gap statistics below); the loader's default is 16
    mgr.timing_stride = int(os.environ["TIMING_STRIDE"])
if os.environ.get("SAMPLER_WAIT", "0") == "1":    # A/B: the fetch stream waits for the sampler's stream on the device, as until round 3
    loader._sampler_done_on_host = False
if os.environ.get("RECORD_STREAM", "0") == "1":   # A/B: the round-3 way of keeping the sampler's tensors alive for the fetch kernels: record_stream on the
    from COALA_GNN.COALA_GNN_DataLoader import _device_tensors, _loader_streams   # fetch stream -- the allocator then records an event per tensor ON that stream at free time
    def _old_keep(batch, ev):
        for t in _device_tensors(batch):
            t.record_stream(_loader_streams("cuda:0")[0])
    loader._keep_until_fetched = _old_keep
pairs = []
def keep(wait):  # keep every (start, end) event pair instead of folding them away
    pairs.extend(mgr._agg_events); mgr._agg_events = []
mgr._fold_events = keep

# ------------------------------------------------------------------ WHY: host intervals, collections, allocator growth, sample completion
host = []        # (name, t0, t1, thread)
gcs = []         # (generation, t0, t1)
reserved = []    # (t, bytes) whenever it changed
sample_done = [] # timing events recorded behind each sample on the sampler's stream
clock = time.perf_counter
if WHY:
    import threading

    def timed(obj, name, label=None):
        fn = getattr(obj, name)
        def wrap(*a, **k):
            t0 = clock()
            try:
                return fn(*a, **k)
            finally:
                host.append((label or name, t0, clock(), threading.get_ident()))
        setattr(obj, name, wrap)
    timed(loader.scheduler, "run", "scheduler.run")
    timed(sampler, "sample_begin"); timed(sampler, "sample_end"); timed(sampler, "sample")
    timed(mgr, "fetch_feature")
    for nm in ("_launch_sample", "_enqueue_fetch", "_produce_one"):
        if hasattr(loader, nm):
            timed(loader, nm)
    _gc_t = {}
    def on_gc(phase, info):
        if phase == "start":
            _gc_t["t"] = clock()
        else:
            gcs.append((info["generation"], _gc_t.get("t", clock()), clock()))
    gc.callbacks.append(on_gc)
    _orig_launch = loader._launch_sample
    def launch_with_event():
        r = _orig_launch()
        e = torch.cuda.Event(enable_timing=True)
        e.record(loader._sample_stream)
        sample_done.append(e)
        return r
    if loader.prefetch <= 0:
        loader._launch_sample = launch_with_event

model = SageMean(dim, 128, 19).cuda(); opt = torch.optim.Adam(model.parameters(), 1e-3, fused=True); lossf = torch.nn.CrossEntropyLoss()
train = os.environ.get("TRAIN", "1") == "1"
gc.collect()
if os.environ.get("GCFREEZE", "1") == "1":
    gc.freeze()   # a full collection of the interpreter inside the loop is a 100 ms hole of its own
torch.cuda.synchronize()
base = torch.cuda.Event(enable_timing=True); base.record(); base.synchronize(); t_base = clock()
t0 = clock(); n = 0
last_res = torch.cuda.memory_reserved()
step_host = []   # host time at which the consumer received step n
step_ev = []     # event on the training stream behind the optimizer step of step n
t_warm = None
for inp, sd, blocks, feat in loader:
    if n == 1200:   # the steady state by the host clock too (the only figure when the fetches carry no event pairs: TIMING_STRIDE)
        torch.cuda.synchronize(); t_warm = clock()
    if WHY:
        step_host.append(clock())
    if train:
        loss = lossf(model(blocks, feat), blocks[-1].dstdata["labels"].view(-1)); opt.zero_grad(); loss.backward(); opt.step()
    if WHY:
        r = torch.cuda.memory_reserved()
        if r != last_res:
            reserved.append((clock(), r - last_res)); last_res = r
        if n % 4 == 0:
            e = torch.cuda.Event(enable_timing=True); e.record(); step_ev.append((n, e))
    n += 1
torch.cuda.synchronize(); dt = clock() - t0
keep(True)
print(f"train={train} prefetch={loader.prefetch} timing_stride={mgr.timing_stride} sampler_wait={not loader._sampler_done_on_host} "
      f"record_stream_on_fetch_stream={os.environ.get('RECORD_STREAM', '0') == '1'}: "
      f"steady state by the host clock (steps 1200..{n}): {(t0 + dt - t_warm) / (n - 1200) * 1e3:.4f} ms/step" if t_warm else "")
if mgr.timing_stride > 1:
    sys.exit(0)
warm = min(1200, max(0, len(pairs) - 400))  # steady state only
from COALA_GNN_Pybind import event_elapsed_ms
def el(a, b):   # ms between two events, torch's or the native handles a fetch carries on its dispatches (valid for 2048 fetches: the steady state fits)
    h = lambda e: e if isinstance(e, int) else int(e.cuda_event)
    return event_elapsed_ms(h(a), h(b), wait=True)
dur = [el(p[0], p[1]) for p in pairs[warm:]]
gap = [el(pairs[i][1], pairs[i + 1][0]) for i in range(warm, len(pairs) - 1)]
print(f"train={train} prefetch={loader.prefetch} refresh_counter={loader.refresh_counter} : {n} steps, {dt / n * 1e3:.3f} ms/step overall (cold start included)")
print(f"  steady state ({len(dur)} fetches): fetch duration mean {st.mean(dur):.4f} ms, median {st.median(dur):.4f} ms; idle gap between fetches mean {st.mean(gap):.4f} ms, "
      f"median {st.median(gap):.4f} ms; sum = {st.mean(dur) + st.mean(gap):.4f} ms/step")
for thr in (0.05, 0.2, 0.5, 1.0):
    sel = [x for x in gap if x > thr]
    print(f"  gaps > {thr:4.2f} ms: {len(sel):5d} ({100 * len(sel) / len(gap):5.1f} %), contributing {sum(sel) / len(gap):.4f} ms/step")
idx = [i for i, x in enumerate(gap) if x > 0.2][:40]
print("  positions (step mod 10) of the first gaps > 0.2 ms:", [(warm + i + 1) % 10 for i in idx])

if WHY:
    # one axis: host seconds.  A GPU event e happened at t_base + base.elapsed_time(e) / 1e3 (the two clocks drift by microseconds over seconds)
    def at(e):
        return t_base + el(base, e) * 1e-3
    fetch_calls = [h for h in host if h[0] == "fetch_feature"]
    print(f"  WHY: {len(gcs)} garbage collections in the loop (by generation: { {k: sum(1 for g_ in gcs if g_[0] == k) for k in (0, 1, 2)} }, "
          f"total {sum(b - a for _, a, b in gcs) * 1e3:.1f} ms), allocator reserved memory changed {len(reserved)} times in the loop "
          f"({sum(1 for t_, _ in reserved if t_ > at(pairs[warm][0]))} of them in the steady state)")
    names = sorted({h[0] for h in host})
    print("  WHY: host calls, mean / p99 / max in ms: " + "; ".join(
        f"{nm} {st.mean([(b - a) * 1e3 for n_, a, b, _ in host if n_ == nm]):.3f} / "
        f"{sorted([(b - a) * 1e3 for n_, a, b, _ in host if n_ == nm])[int(0.99 * (sum(1 for h in host if h[0] == nm) - 1))]:.3f} / "
        f"{max([(b - a) * 1e3 for n_, a, b, _ in host if n_ == nm]):.3f}" for nm in names))
    big = [i for i, x in enumerate(gap) if x > 0.3]
    print(f"  WHY: the {len(big)} gaps > 0.3 ms, one per line: fetch#, gap ms | when fetch t+1 was ENQUEUED relative to the end of fetch t (ms; >0 = the host was late) | "
          f"sample completion relative to the end of fetch t | host calls / collections / allocator growth overlapping the gap")
    for i in big[:60]:
        k = warm + i                       # gap between fetch k and k+1
        g0, g1 = at(pairs[k][1]), at(pairs[k + 1][0])
        enq = fetch_calls[k + 1][1] - g0 if k + 1 < len(fetch_calls) else float("nan")
        smp = (at(sample_done[k + 1]) - g0) * 1e3 if k + 1 < len(sample_done) else float("nan")
        over = [(nm, (min(b, g1) - max(a, g0)) * 1e3, (b - a) * 1e3) for nm, a, b, _ in host if b > g0 and a < g1 and nm != "_produce_one"]
        over = [f"{nm} {o:.2f}/{d:.2f}" for nm, o, d in over if o > 0.05]
        gg = [f"gc{gen} {(b - a) * 1e3:.2f}" for gen, a, b in gcs if b > g0 - 0.002 and a < g1]
        rr = [f"reserved{d / 1e6:+.0f}MB" for t_, d in reserved if g0 - 0.005 < t_ < g1 + 0.005]
        print(f"    fetch {k:5d}  gap {gap[i]:7.3f} | enqueued {enq * 1e3:+8.3f} | sample done {smp:+8.3f} | {' '.join(over + gg + rr) or '-'}")
