#!/usr/bin/env python3
"""probe_gather_kernel IN SITU: the default bench workload (IGB-medium shape, fan-out 5,5, isolated 4 GiB cache, pinned-host cold
tier), one cache handle per variant (COALA_K1_* environment knobs are read at handle creation), 400 warm-up minibatches, then
the kernel's average duration over 200 timed minibatches by HIP events attached to its launches (COALA_FLAG_PROFILE) -- i.e. with the cold fill of the
previous step in front of every launch, which a micro-benchmark over one repeated batch (round 2: profiles/r02_k1_variants.txt) does not have.

  python tools/k1_insitu.py "GRID=2048" "GRID=1024" "GRID=2048,PASSES=2" ...      (knob list per variant, comma separated)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("coala_build", os.path.join(ROOT, "coala-gnn_amd", "build.py"))
_bm = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_bm)
if not os.path.exists(_bm.DEV_LIB_PATH) or os.path.getmtime(_bm.DEV_LIB_PATH) < max(os.path.getmtime(p) for p in _bm.sources() + _bm.HEADERS):
    _bm.build_lib(dev=True)
os.environ["COALA_HIP_LIB"] = os.environ.get("K1_LIB", _bm.DEV_LIB_PATH)   # the development build: launch-geometry knobs from the environment
import torch  # noqa: E402
import COALA_GNN_Pybind as P  # noqa: E402
from COALA_GNN.synthetic import PinnedFeatureTable, fill_table, powerlaw_csc  # noqa: E402
from COALA_GNN.sampler import NeighborSampler  # noqa: E402

# default: BASELINE configs[1] (IGB-medium 5,5, 4 GiB).  configs[3]'s shape (papers100M: 512-B lines, 16 GiB cache = a 134 MB tag table,
# 289 k rows per minibatch at 62 % hits):  ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6
rows, dim, batch, cache_mb = int(os.environ.get("ROWS", 10_000_000)), int(os.environ.get("DIM", 1024)), 1024, int(os.environ.get("CACHE_MB", 4096))
fanout = [int(f) for f in os.environ.get("FANOUT", "5,5").split(",")]
max_rows = batch
for f in fanout:
    max_rows *= f + 1
torch.cuda.set_device(0)
t0 = time.time()
table = PinnedFeatureTable(rows, dim, 0)
fill_table(table.cpu_tensor, 0, device="cuda:0")
indptr, indices = powerlaw_csc(rows, float(os.environ.get("DEG", 12.0)), seed=0, device="cuda:0")
train_ids = torch.randperm(int(0.6 * rows), generator=torch.Generator().manual_seed(0))
sampler = NeighborSampler(fanout, seed=0)
graph = sampler.make_graph(indptr, indices)
batches = [sampler.sample(graph, train_ids[s * batch: (s + 1) * batch].cuda(), step=s)[0] for s in range(620)]
torch.cuda.synchronize()
print(f"# setup {time.time() - t0:.1f}s; rows per minibatch ~{sum(b.numel() for b in batches[400:]) / 220:.0f}", flush=True)
ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
if "--stages" in sys.argv:
    # where the fixed cost goes: K1's dependency chain cut after each link (development switch of the kernel itself: any line size
    # and tag width), launched right behind a real step's cold fill, on K1's own grid
    import ctypes as C
    from COALA_GNN_Pybind import current_stream
    L = C.CDLL(os.environ["COALA_HIP_LIB"])
    L.coala_dev_k1_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    cache = P.Isolated_Cache(ctrl, None, 0, 1, cache_mb, table.device_ptr, num_rows=rows, sync=False, max_batch=max_rows)
    out = torch.empty((max_rows, dim), dtype=torch.float32, device="cuda")
    for b in batches[:420]:
        cache.read_feature(out.data_ptr(), b.data_ptr(), b.numel())
    torch.cuda.synchronize()
    names = {-1: "empty event bracket", 0: "empty kernel on K1's grid (launch + drain)", 1: "+ ids", 2: "+ tag sets, ballots", 3: "+ line loads of the hit rows (no stores)",
             4: "product kernel without miss bookkeeping", 5: "product kernel"}
    print(f"# dim {dim}, tags {cache.geometry().tag_set_bytes} B/set, cache {cache_mb} MiB; separate hipEvent brackets (each contains the ~4.5 us of the empty bracket)")
    for stage in (-1, 0, 1, 2, 3, 4, 5, 4, 5):
        evs = []
        for k, b in enumerate(batches[420:619]):
            cache.read_feature(out.data_ptr(), b.data_ptr(), b.numel())      # a real step: K1 + the PCIe-bound K2
            nxt = batches[421 + k]
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            if stage >= 0:
                L.coala_dev_k1_stage(cache._h, out.data_ptr(), nxt.data_ptr(), nxt.numel(), stage, current_stream())
            e.record()
            evs.append((a, e))
        torch.cuda.synchronize()
        us = sorted(x.elapsed_time(y) * 1e3 for x, y in evs)
        print(f"stage {stage:2d} {names[stage]:45s} mean {sum(us) / len(us):6.2f} us   median {us[len(us) // 2]:6.2f} us", flush=True)
    sys.exit(0)
variants = [a for a in sys.argv[1:] if not a.startswith("--")] or ["GRID=2048"]
# OUT_BUFFERS=n: the fetches rotate over n output buffers.  bench.py and the loaders get a fresh tensor from the manager for every fetch (two
# or three blocks of torch's allocator in rotation); with ONE buffer the rows of step s may still sit in the 256 MiB Infinity Cache when step
# s+1 overwrites them, which flatters the kernel
outs = [torch.empty((max_rows, dim), dtype=torch.float32, device="cuda") for _ in range(int(os.environ.get("OUT_BUFFERS", 3)))]
out = outs[0]
for rep in range(int(os.environ.get("REPS", 2))):
    for v in variants:
        for k in list(os.environ):
            if k.startswith("COALA_K1_") or k.startswith("COALA_K2_"):
                del os.environ[k]
        for kv in v.split(","):
            if kv:
                k, val = kv.split("=")
                os.environ[("COALA_" if k.startswith("K2_") else "COALA_K1_") + k] = val   # K2_GRID=24, K2_TILE_ROWS=32: the cold fill's knobs
        cache = P.Isolated_Cache(ctrl, None, 0, 1, cache_mb, table.device_ptr, num_rows=rows, profile=True, sync=False, max_batch=max_rows)
        tagb = cache.geometry().tag_set_bytes
        for b in batches[:420]:
            cache.read_feature(out.data_ptr(), b.data_ptr(), b.numel())
        torch.cuda.synchronize()
        cache.stats(reset=True)
        cache.profile(reset=True)
        pre = os.environ.get("COALA_K1_PRE", "")   # experiment: a wide kernel right in front of every K1 (clock / power-state probe)
        scratch = torch.empty(int(pre) << 18, dtype=torch.float32, device="cuda") if pre else None
        # BUSY=n: n compute-bound kernels (bf16 2048^3 matrix products, ~15 us each, 24 MB of operands) between the PCIe-bound fill of one step and the
        # K1 of the next: does K1 start on a chip that has clocked down during ~1 ms of near-idle?  (events attached to K1 itself: the products are not timed)
        busy = int(os.environ.get("COALA_K1_BUSY", "0"))
        if busy:
            mm_a = torch.randn((2048, 2048), dtype=torch.bfloat16, device="cuda")
            mm_b = torch.randn((2048, 2048), dtype=torch.bfloat16, device="cuda")
            mm_c = torch.empty((2048, 2048), dtype=torch.bfloat16, device="cuda")
        t1 = time.perf_counter()
        seg_us = []      # K1 per quarter of the timed minibatches: does the figure drift within ONE handle, or only from handle to handle?
        acc = None
        for k, b in enumerate(batches[420:]):
            if scratch is not None:
                scratch.zero_()
            for _ in range(busy):
                torch.mm(mm_a, mm_b, out=mm_c)
            cache.read_feature(outs[k % len(outs)].data_ptr(), b.data_ptr(), b.numel())
            if os.environ.get("SEGMENTS") and (k + 1) % 50 == 0:
                torch.cuda.synchronize()
                q = cache.profile()
                prev = acc or (0.0, 0)
                seg_us.append((q.gather_ms - prev[0]) / max(q.gather_launches - prev[1], 1) * 1e3)
                acc = (q.gather_ms, q.gather_launches)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t1) / 200 * 1e3
        p = cache.profile()
        if seg_us:
            print("   K1 per 50 minibatches: " + " ".join(f"{u:6.2f}" for u in seg_us), flush=True)
        hit, miss, _ = cache.stats()
        alg = p.gather_rows * (8 + tagb) + p.gather_hits * 2 * dim * 4
        us = p.gather_ms / p.gather_launches * 1e3
        print(f"{v:40s} tags {tagb} B/set  K1 {us:7.2f} us (events attached to the launch)  {alg / p.gather_launches / us / 1e3:7.1f} GB/s = "
              f"{alg / p.gather_launches / us / 1e3 / 80:5.1f} % of 8 TB/s   K2 {p.fill_ms / p.fill_launches * 1e3:8.1f} us   step {wall:.4f} ms   hit {hit / (hit + miss):.4f}", flush=True)
        if os.environ.get("ALLHIT"):   # the BASELINE section 4 micro-benchmark on the same handle: 36,864 unique ids, every row a hit
            ids = torch.randperm(rows, device="cuda", generator=torch.Generator(device="cuda").manual_seed(12345))[:max_rows]
            for _ in range(3):
                cache.read_feature(out.data_ptr(), ids.data_ptr(), ids.numel())
            torch.cuda.synchronize()
            cache.stats(reset=True)
            cache.profile(reset=True)
            for k in range(100):
                cache.read_feature(outs[k % len(outs)].data_ptr(), ids.data_ptr(), ids.numel())
            torch.cuda.synchronize()
            p = cache.profile()
            alg = p.gather_rows * (8 + tagb) + p.gather_hits * 2 * dim * 4
            us = p.gather_ms / p.gather_launches * 1e3
            print(f"{'  all-hit, max_sample rows':40s} K1 {us:7.2f} us   {alg / p.gather_launches / us / 1e3:7.1f} GB/s = {alg / p.gather_launches / us / 1e3 / 80:5.1f} % of 8 TB/s", flush=True)
        cache.close()
