#!/usr/bin/env python3
"""The hand-over between the cold fill of one minibatch and the probe of the next on ONE stream, by itself: the default workload's minibatches through
cache.read_feature in a loop, nothing else on the GPU.  Run under `rocprofv3 --kernel-trace` and read with tools/gap_trace_analyze.py.  Variants:
  VARIANT=plain    plain launches, no event anywhere
  VARIANT=events   coala_cache_fetch_events: begin event on K1's launch, end event on K2's
  VARIANT=profile  COALA_FLAG_PROFILE: begin AND end event on both launches
  VARIANT=records  plain launches + the round-3 packets: a timing pair and a completion event recorded around every read
  VARIANT=events_wait   as events, and a consumer stream waits for every end event (coala_stream_wait_event), as the loader's consumer does
  VARIANT=one_record[_wait]  plain launches + ONE event recorded behind every read (no timing pair) [and a consumer stream waits for it]
  STREAM=side|default   a created stream or the default one
Development tool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
import torch  # noqa: E402
import COALA_GNN_Pybind as P  # noqa: E402
from COALA_GNN.sampler import NeighborSampler  # noqa: E402
from COALA_GNN.synthetic import PinnedFeatureTable, fill_table, powerlaw_csc  # noqa: E402

rows, dim, batch, cache_mb = 10_000_000, 1024, 1024, 4096
variant, which = os.environ.get("VARIANT", "plain"), os.environ.get("STREAM", "side")
torch.cuda.set_device(0)
table = PinnedFeatureTable(rows, dim, 0)
fill_table(table.cpu_tensor, 0, device="cuda:0")
indptr, indices = powerlaw_csc(rows, 12.0, seed=0, device="cuda:0")
train_ids = torch.randperm(int(0.6 * rows), generator=torch.Generator().manual_seed(0))
sampler = NeighborSampler([5, 5], seed=0)
graph = sampler.make_graph(indptr, indices)
batches = [sampler.sample(graph, train_ids[s * batch: (s + 1) * batch].cuda(), step=s)[0] for s in range(800)]
torch.cuda.synchronize()
ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
cache = P.Isolated_Cache(ctrl, None, 0, 1, cache_mb, table.device_ptr, num_rows=rows, profile=variant == "profile", sync=False, max_batch=batch * 36)
if variant.startswith("events"):
    cache.fetch_events(True)
consumer = torch.cuda.Stream()
outs = [torch.empty((batch * 36, dim), dtype=torch.float32, device="cuda") for _ in range(3)]
stream = torch.cuda.Stream() if which == "side" else torch.cuda.current_stream()
keep = []
with torch.cuda.stream(stream):
    for k, b in enumerate(batches):
        if variant == "records":
            e0, e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
            e0.record()
        cache.read_feature(outs[k % 3].data_ptr(), b.data_ptr(), b.numel())
        if variant == "records":
            e1.record(); e2.record()
            keep.append((e0, e1, e2))
        elif variant == "events_wait":
            P.stream_wait_event(cache.last_fetch_events()[1], int(consumer.cuda_stream))
        elif variant.startswith("one_record"):
            e = torch.cuda.Event()
            e.record()
            if variant.endswith("_wait"):
                consumer.wait_event(e)
            keep.append(e)
        if len(keep) > 64:
            keep.pop(0)
torch.cuda.synchronize()
print(f"VARIANT={variant} STREAM={which}: {len(batches)} reads done", flush=True)
