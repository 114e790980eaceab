#!/usr/bin/env python3
"""What the packets of a distributed fetch cost on the caller's stream.  G logical ranks (host threads) on ONE GPU, owner-partitioned cache and
cold tier, the fused native bucketed fetch over the in-process transport with the count exchange issued one step ahead -- the sequence
bench.py runs at N > 1 -- on pre-sampled minibatches, each rank on its own stream with a consumer stream behind it.  Two arms, alternating:

  records  the round-3 form: the hand-over event of every fill round recorded behind its kernel, every row round waited for on the caller's
           stream (COALA_COMM_PLAIN_EVENTS=1, development build), plus what the manager and the loader added per fetch: a timing pair and a
           completion event recorded on the stream, which the consumer's stream waits for
  riding   round 4: the hand-over events ride on the fill launches, the caller's stream waits for the last row round only, and the fetch's begin
           / end events ride on its probe / last fill (coala_comm_fetch_events); the consumer's stream waits for the two end events

Times are per fetch by the host clock over all ranks (one GPU, one PCIe link and one copy kernel for the exchange: the ABSOLUTE step time does not
transfer to N GPUs; the difference between the arms is packets on the stream, which does).  Development tool:  python tools/dist_packets_probe.py [--ranks 2]"""
import argparse
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "coala-gnn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("coala_build", os.path.join(ROOT, "coala-gnn_amd", "build.py"))
_bm = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_bm)
os.environ["COALA_HIP_LIB"] = _bm.build_lib(dev=True)     # the arms differ by a development knob of the communicator
import torch  # noqa: E402
import COALA_GNN_Pybind as P  # noqa: E402
from COALA_GNN.COALA_GNN_Manager import NativeExchange  # noqa: E402
from COALA_GNN.sampler import NeighborSampler  # noqa: E402
from COALA_GNN.synthetic import PinnedFeatureTable, feature_rows_torch, fill_table_partition, powerlaw_csc  # noqa: E402
from COALA_GNN_Pybind import _capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--cache-mb", type=int, default=1600)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--measure-from", type=int, default=200)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    G, dim, fan, batch = a.ranks, a.dim, [5, 5], 1024
    torch.cuda.set_device(0)
    L = _capi.load()
    tables = []
    for r in range(G):
        t = PinnedFeatureTable((a.rows + G - 1) // G, dim, 0)
        fill_table_partition(t.cpu_tensor, 0, r, G, device="cuda:0")
        tables.append(t)
    indptr, indices = powerlaw_csc(a.rows, 12.0, seed=0, device="cuda")
    train = torch.randperm(int(0.6 * a.rows), generator=torch.Generator().manual_seed(0))
    ctrl = P.SSD_GNN_SSD_Controllers(1, dim * 4, 1024, 0, 0, dim, True)
    smp = NeighborSampler(fan, seed=0, bucket_by_owner=G)
    graph = smp.make_graph(indptr, indices)
    # pre-sampled: the sampler is not what is measured
    batches = [[smp.sample(graph, train[((s * G + r) * batch): ((s * G + r) + 1) * batch].cuda(), step=s) for s in range(a.steps)] for r in range(G)]
    torch.cuda.synchronize()
    print(f"# {G} logical ranks on one GPU, {a.rows} x {dim} table partitioned by owner, {a.cache_mb} MiB of cache per rank, fan-out 5,5 bs 1024 "
          f"(~{sum(b[0].numel() for b in batches[0]) / a.steps:.0f} rows per fetch), in-process transport, counts issued one step ahead; "
          f"steps {a.measure_from}..{a.steps} timed", flush=True)
    for rep in range(a.reps):
        for arm in ("records", "riding"):
            os.environ["COALA_COMM_PLAIN_EVENTS"] = "1" if arm == "records" else "0"
            caches = [P.Isolated_Cache(ctrl, None, r, G, a.cache_mb, tables[r].device_ptr, num_rows=a.rows, rank=r, sync=False, cold_partitioned=True,
                                       max_batch=batch * 36 * G) for r in range(G)]
            group = C.c_void_p()
            _capi.check(L.coala_comm_group_create(G, C.byref(group)))
            exs = [NativeExchange(None, 0, r, G, 0, inproc_group=group) for r in range(G)]
            res, errors = [None] * G, []
            bar = threading.Barrier(G, timeout=300)

            def worker(r):
                try:
                    torch.cuda.set_device(0)
                    stream, side, consumer = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
                    ex, cache = exs[r], caches[r]
                    if arm == "riding":
                        ex.fetch_events(1)
                    my = batches[r]
                    keep = []
                    with torch.cuda.stream(side):
                        tk = ex.counts_begin(my[0][2][0].owner_counts.data_ptr())
                    with torch.cuda.stream(stream):
                        for s in range(a.steps):
                            if s == a.measure_from:
                                stream.synchronize(); consumer.synchronize()
                                bar.wait()
                                t0 = time.perf_counter()
                            ids, _, blocks = my[s]
                            tk_now = tk
                            if s + 1 < a.steps:
                                with torch.cuda.stream(side):
                                    tk = ex.counts_begin(my[s + 1][2][0].owner_counts.data_ptr())
                            n = ids.numel()
                            feat = torch.empty((n, dim), dtype=torch.float32, device="cuda")
                            if arm == "records":
                                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                                e0.record()
                                ex.fetch_bucketed(cache, feat.data_ptr(), ids.data_ptr(), n, blocks[0].owner_counts.data_ptr(), ticket=tk_now)
                                e1.record()
                                done = torch.cuda.Event()
                                done.record()
                                consumer.wait_event(done)
                            else:
                                ex.fetch_bucketed(cache, feat.data_ptr(), ids.data_ptr(), n, blocks[0].owner_counts.data_ptr(), ticket=tk_now)
                                _, end_st, end_cs = ex.last_fetch_events()
                                for h in (end_st, end_cs):
                                    if h:
                                        P.stream_wait_event(h, int(consumer.cuda_stream))
                            feat.record_stream(consumer)
                            keep.append(feat)
                            if len(keep) > 3:
                                keep.pop(0)
                            if s == a.steps - 1:
                                with torch.cuda.stream(consumer):
                                    ok = bool(torch.equal(feat, feature_rows_torch(ids, dim, 0)))
                        stream.synchronize(); consumer.synchronize()
                        bar.wait()
                    res[r] = ((time.perf_counter() - t0) / (a.steps - a.measure_from) * 1e3, ok)
                except BaseException as e:  # noqa: BLE001
                    errors.append((r, repr(e)))
                    bar.abort()
            ts = [threading.Thread(target=worker, args=(r,)) for r in range(G)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            if errors:
                raise SystemExit(f"{arm}: {errors}")
            assert all(ok for _, ok in res), "the consumer's stream saw rows that differ from the table"
            hit = sum(c.stats()[0] for c in caches); miss = sum(c.stats()[1] for c in caches)
            print(f"rep {rep} {arm:8s} {max(ms for ms, _ in res):8.4f} ms per fetch (slowest rank; all ranks: {' '.join(f'{ms:.4f}' for ms, _ in res)})   "
                  f"hit ratio {hit / max(hit + miss, 1):.4f}   last minibatch bit-exact on the consumer's stream", flush=True)
            for e in exs:
                e.close()
            _capi.check(L.coala_comm_group_destroy(group))
            for c in caches:
                c.close()


if __name__ == "__main__":
    main()
