"""fetch_feature orchestration: isolated cache, or the owner-partitioned cache with RCCL all-to-all-v over xGMI.

Mirror of COALA-GNN-Setup/COALA_GNN/COALA_GNN_Manager.py (reference): COALA_GNN_Manager :44-230 with the same constructor
and methods; NVShmem_Tensor_Manager :8-40 becomes a thin holder of torch-owned HBM buffers (no cupy, no symmetric heap).

Backends (reference strings kept):
  "isolated" -- each GPU caches what it touches: one call into the probe+gather / rank / fill kernels.
  "nccl"     -- owner = id % G; ids and rows move with torch.distributed all_to_all_single (RCCL on GPU tensors).
  "nvshmem"  -- same partitioning; NVSHMEM's device-initiated puts do not exist here, so it rides the same exchange
                through SSD_GNN_NVSHMEM_Cache.send_requests / read_feature (the reference's call sequence, :131-132).
Per minibatch the exchange is: route (stable bucketing on the GPU) -> all-to-all of G counts -> ONE host read of the
2G counts -> all-to-all-v of ids (exact sizes) -> owner serve (one batch: concatenation in source-rank order) ->
all-to-all-v of rows -> un-permute.  The reference instead moves full-capacity id buffers (:159-163) and G(G-1) serial
send/recv pairs (:194-203).
"""
import contextlib
import time

import torch
import torch.distributed as dist

from COALA_GNN_Pybind import NVSHMEM_Manager, SSD_GNN_SSD_Controllers, SSD_GNN_NVSHMEM_Cache, Isolated_Cache

__all__ = ["COALA_GNN_Manager", "NVShmem_Tensor_Manager", "AllToAllExchange", "NativeExchange"]


class NVShmem_Tensor_Manager(object):
    """COALA_GNN_Manager.py:8-40.  Hands out the per-step output tensor and owns the request buffer (plain HBM).

    The reference returns a view of ONE symmetric buffer every step (:127-128): the tensor of step t is overwritten by the
    fetch of step t+1.  Here every step gets its own tensor from torch's caching allocator, like the other backends: a
    prefetching loader fetches step t+k on a side stream while the training kernels of step t still read their rows, and only
    the allocator's stream bookkeeping (record_stream, done by the loader) makes that reuse safe -- a persistent ring is not."""

    def __init__(self, max_rows, dim, n_gpus, device, ring=2):
        self.device = device
        self.dim = dim
        self.index_tensor = torch.empty((n_gpus, max_rows * 2), dtype=torch.int64, device=device)

    def get_batch_tensor(self, shape):
        return torch.empty((int(shape[0]), self.dim), dtype=torch.float32, device=self.device)

    def get_index_tensor(self):
        return self.index_tensor

    def get_index_tensor_ptr(self):
        return self.index_tensor.data_ptr()


def _round_slices(cnt, dis, rounds, skip):
    """[(begin, end)] per round: slice k of every peer's segment [dis[p], dis[p] + cnt[p]), peer `skip` left out."""
    out = []
    for k in range(rounds):
        out.append([(dis[p] + cnt[p] * k // rounds, dis[p] + cnt[p] * (k + 1) // rounds) if p != skip else (dis[p], dis[p])
                    for p in range(len(cnt))])
    return out


class AllToAllExchange(object):
    """ids out / rows back over one torch.distributed process group -- the same split-phase sequence as the native call
    (coala_comm.cpp): route -> counts -> ONE host read -> ids -> probe (the requester's own shard goes straight into the
    output tensor) -> per round { cold fill of slice k of every peer's segment | rows of slice k on a side stream } ->
    un-permute per round.  `ops` provides the device primitives as pointer-level calls: a COALA_GNN_Pybind cache object in
    the product; tests may inject another provider to exercise this host logic with gloo on CPU tensors."""

    def __init__(self, group, rank, world, dim, device, stage_through_host=None, rounds=None):
        self.group, self.rank, self.world, self.dim, self.device = group, rank, world, dim, device
        # RCCL moves GPU tensors directly.  A gloo group cannot (no CUDA all-to-all): then the buffers are staged through
        # host memory -- transport only, used by the multi-process tests on a one-GPU box; the kernels stay on the GPU.
        on_gpu = str(device).startswith("cuda")
        if stage_through_host is None:
            stage_through_host = (world > 1 and on_gpu and dist.is_initialized() and dist.get_backend(group) == "gloo")
        self.stage_through_host = bool(stage_through_host)
        if rounds is None:  # the same default and the same switch as the native call (coala_comm.cpp)
            import os
            rounds = os.environ.get("COALA_EXCHANGE_ROUNDS", "2")
        self.rounds = min(max(1, int(rounds)), 8)
        self._pending = None
        self.counts = torch.zeros(world, dtype=torch.int64, device=device)
        self.offsets = torch.zeros(world + 1, dtype=torch.int64, device=device)
        self.recv_counts = torch.zeros(world, dtype=torch.int64, device=device)
        self._both_dev = torch.zeros((2, world), dtype=torch.int64, device=device)
        # pinned: a D2H copy into pageable memory is staged by the runtime and can wait on more than this stream
        self._both_host = torch.zeros((2, world), dtype=torch.int64, pin_memory=on_gpu)
        # persistent workspaces, grown on demand (no per-step allocation)
        self._ws = {}
        # (high priority = a hardware queue of its own, as the native call's communication stream)
        self._side = torch.cuda.Stream(device=device, priority=-1) if (on_gpu and world > 1 and not self.stage_through_host) else None
        self.last_send_counts = None
        self.last_recv_counts = None
        # profile = True brackets the row exchange with HIP events (bench.py's xGMI figure: BASELINE.md "achieved_xGMI")
        self.profile = False
        self._row_events = []
        self.rows_a2a_ms = 0.0
        self.rows_a2a_calls = 0
        self.remote_rows_in = 0

    def _buf(self, name, numel, dtype):
        t = self._ws.get(name)
        if t is None or t.numel() < numel:
            if t is not None and t.is_cuda:
                torch.cuda.current_stream().synchronize()  # kernels of the previous step may still use the old buffer
            t = torch.empty(max(int(numel * 1.25), 1024), dtype=dtype, device=self.device)
            self._ws[name] = t
        return t

    def fold_profile(self):
        """Finish the pending event pairs; -> (milliseconds in the row exchange, calls, rows received from other ranks)."""
        for a, b in self._row_events:
            b.synchronize()
            self.rows_a2a_ms += a.elapsed_time(b)
        self._row_events = []
        return self.rows_a2a_ms, self.rows_a2a_calls, self.remote_rows_in

    def reset_profile(self):
        self.fold_profile()
        self.rows_a2a_ms, self.rows_a2a_calls, self.remote_rows_in = 0.0, 0, 0

    def _a2a(self, out, inp, out_splits=None, in_splits=None):
        if self.world == 1:
            out.copy_(inp)
        elif self.stage_through_host:
            h_out = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(h_out, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(h_out)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    def _a2a_slices(self, dst, dst_slices, src, src_slices):
        """Every peer p receives src[src_slices[p]] and delivers into dst[dst_slices[p]] (rows; slices are (begin, end))."""
        outs = [dst[b:e] for b, e in dst_slices]
        ins = [src[b:e] for b, e in src_slices]
        if self.stage_through_host or not dst.is_cuda:
            h_in = torch.cat([t.cpu() for t in ins]) if ins else src[:0].cpu()
            h_out = torch.empty((sum(t.shape[0] for t in outs),) + tuple(dst.shape[1:]), dtype=dst.dtype)
            dist.all_to_all_single(h_out, h_in, [t.shape[0] for t in outs], [t.shape[0] for t in ins], group=self.group)
            pos = 0
            for t in outs:
                t.copy_(h_out[pos: pos + t.shape[0]])
                pos += t.shape[0]
        else:
            dist.all_to_all(outs, ins, group=self.group)

    def send_requests(self, ops, idx_ptr, n, req_ptr, max_index):
        """route + count exchange + id exchange.  ssd_gnn_cache.cuh:111-129 / COALA_GNN_Manager.py:152-165."""
        n = int(n)
        node = self._buf("node", n, torch.int64)
        mp = self._buf("map", n, torch.int64)
        ops.route(idx_ptr, n, self.world, node.data_ptr(), mp.data_ptr(), self.counts.data_ptr(), self.offsets.data_ptr(), 0)
        self._a2a(self.recv_counts, self.counts)
        self._both_dev[0].copy_(self.counts)
        self._both_dev[1].copy_(self.recv_counts)
        self._both_host.copy_(self._both_dev, non_blocking=True)
        if self._both_dev.is_cuda:
            torch.cuda.current_stream().synchronize()  # the one host synchronisation of the step
        send_c, recv_c = self._both_host[0].tolist(), self._both_host[1].tolist()
        total_recv = int(sum(recv_c))
        recv_ids = self._buf("recv_ids", total_recv, torch.int64)
        self._a2a(recv_ids[:total_recv], node[:n], recv_c, send_c)
        self._pending = (n, node, mp, send_c, recv_c, recv_ids, total_recv)
        self.last_send_counts, self.last_recv_counts = send_c, recv_c

    def _serve_in_rounds(self, ops, n, send_c, recv_c, recv_ids, own_out_ptr, own_map_ptr, rows_dst):
        """Owner and requester side of one step after the ids have arrived: probe the ONE batch (own segment redirected to
        own_out_ptr through own_map_ptr), then per round {cold fill of slice k of every peer's segment | rows of slice k into
        rows_dst on the side stream}.  -> (requester-side slices per round, events to wait for per round)."""
        G, me, dim = self.world, self.rank, self.dim
        total_recv = int(sum(recv_c))
        sdis = [sum(send_c[:p]) for p in range(G)]
        rdis = [sum(recv_c[:p]) for p in range(G)]
        rows_send = self._buf("rows_send", total_recv * dim, torch.float32)[: max(total_recv, 1) * dim].view(-1, dim)
        K = 1 if G == 1 else self.rounds
        if total_recv:
            # one batch per owner and step: the concatenation in source-rank order; the own segment lands in the caller's tensor
            ops.serve_probe_redirect(rows_send.data_ptr(), recv_ids.data_ptr(), total_recv, rdis[me], rdis[me] + recv_c[me],
                                     own_out_ptr, own_map_ptr)
        fill = _round_slices(recv_c, rdis, K, me)      # owner side: positions of the batch / rows of rows_send
        land = _round_slices(send_c, sdis, K, me)      # requester side: rows of rows_dst (bucket order)
        cur = torch.cuda.current_stream() if rows_send.is_cuda else None
        timed = self.profile and rows_send.is_cuda and G > 1
        ev_t = None
        done = []
        for k in range(K):
            if total_recv:
                rng = [r for r in fill[k] if r[1] > r[0]]
                if k == K - 1 and recv_c[me]:
                    rng.append((rdis[me], rdis[me] + recv_c[me]))  # own segment: nobody waits for it on a link
                ops.serve_fill_ranges(rows_send.data_ptr(), recv_ids.data_ptr(), total_recv, rng)
            if G == 1:
                continue   # (every rank of a larger group takes part in every round, whatever its own batch size: it is a collective)
            if self._side is not None:
                ev = torch.cuda.Event()
                ev.record(cur)
                self._side.wait_event(ev)
            with torch.cuda.stream(self._side) if self._side is not None else contextlib.nullcontext():
                if timed and k == 0:
                    ev_t = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev_t[0].record()
                self._a2a_slices(rows_dst if rows_dst is not None else rows_send[:0], land[k], rows_send, fill[k])
                if timed and k == K - 1:
                    ev_t[1].record()
                e2 = None
                if self._side is not None:
                    e2 = torch.cuda.Event()
                    e2.record()
                done.append(e2)
        if timed and ev_t is not None:
            self._row_events.append(ev_t)
            self.rows_a2a_calls += 1
            self.remote_rows_in += n - int(send_c[me])
        return land, done, cur, sdis

    def read_feature(self, ops, out_ptr, req_ptr, max_index):
        """probe + rounds of {fill, row exchange} + un-permute.  ssd_gnn_cache.cuh:132-174 / COALA_GNN_Manager.py:167-209."""
        n, node, mp, send_c, recv_c, recv_ids, total_recv = self._pending
        self._pending = None
        me = self.rank
        rows_recv = self._buf("rows_recv", n * self.dim, torch.float32)[: max(n, 1) * self.dim].view(-1, self.dim)
        sdis_me = sum(send_c[:me])
        land, done, cur, _ = self._serve_in_rounds(ops, n, send_c, recv_c, recv_ids, out_ptr, mp.data_ptr() + sdis_me * 8, rows_recv)
        for k, e2 in enumerate(done):  # un-permute round by round as the rows arrive
            if e2 is not None:
                cur.wait_event(e2)
            ops.scatter_ranges(out_ptr, rows_recv.data_ptr(), mp.data_ptr(), [r for r in land[k] if r[1] > r[0]])

    def fetch(self, ops, out_ptr, idx_ptr, n, max_index=0):
        self.send_requests(ops, idx_ptr, n, 0, max_index)
        self.read_feature(ops, out_ptr, 0, max_index)

    def fetch_bucketed(self, ops, out_ptr, idx_ptr, n, counts_ptr):
        """idx already bucketed by owner (NeighborSampler(bucket_by_owner=G)), counts on the device: no routing pass, every
        peer's rows are received straight into the output tensor, nothing to un-permute (mirror of
        coala_cache_fetch_distributed_bucketed)."""
        n, G, me, dim = int(n), self.world, self.rank, self.dim
        self.counts.copy_(_tensor_view(counts_ptr, G, torch.int64, self.counts))
        self._a2a(self.recv_counts, self.counts)
        self._both_dev[0].copy_(self.counts)
        self._both_dev[1].copy_(self.recv_counts)
        self._both_host.copy_(self._both_dev, non_blocking=True)
        if self._both_dev.is_cuda:
            torch.cuda.current_stream().synchronize()
        send_c, recv_c = self._both_host[0].tolist(), self._both_host[1].tolist()
        if sum(send_c) != n:
            raise RuntimeError(f"the bucket counts sum to {sum(send_c)} for a batch of {n} ids")
        self.last_send_counts, self.last_recv_counts = send_c, recv_c
        total_recv = int(sum(recv_c))
        node = _tensor_view(idx_ptr, n, torch.int64, self.counts)
        out = _tensor_view(out_ptr, n * dim, torch.float32, self.counts).view(-1, dim) if n else None
        recv_ids = self._buf("recv_ids", total_recv, torch.int64)
        self._a2a(recv_ids[:total_recv], node[:n], recv_c, send_c)
        # bucket order IS the caller's order: the own bucket sits at its offset of `out`, every peer's rows land at theirs
        own_out = int(out_ptr) + sum(send_c[:me]) * dim * 4
        _, done, cur, _ = self._serve_in_rounds(ops, n, send_c, recv_c, recv_ids, own_out, 0, out)
        for e2 in done:
            if e2 is not None:
                cur.wait_event(e2)


def _tensor_view(ptr, numel, dtype, like):
    """A tensor over caller-owned memory at `ptr` on the device of `like` (CPU tensors for the gloo tests, HBM otherwise)."""
    if numel == 0:
        return torch.empty(0, dtype=dtype, device=like.device)
    if like.is_cuda:
        from .Shared_Tensor import tensor_from_pointer
        return tensor_from_pointer(int(ptr), (int(numel),), dtype, like.device)
    import ctypes
    import numpy as np
    npdt = np.dtype(torch.empty(0, dtype=dtype).numpy().dtype)
    buf = (ctypes.c_char * (int(numel) * npdt.itemsize)).from_address(int(ptr))
    return torch.from_numpy(np.frombuffer(buf, dtype=npdt))


class NativeExchange(object):
    """The same exchange as ONE native call (coala_cache_fetch_distributed: route, counts, ids, probe with the own shard
    delivered in place, fill rounds overlapped with row rounds on the communicator's own HIP stream, un-permute -- all
    inside libcoala_hip.so).  Own RCCL communicator per cache group, bootstrapped by broadcasting the 128-byte ncclUniqueId
    over a torch.distributed CPU group; or, with inproc_group=, a rank of an in-process group (host threads)."""

    def __init__(self, bootstrap_group, src_global_rank, rank, world, device_index, inproc_group=None, rounds=None):
        import ctypes as C
        from COALA_GNN_Pybind import _capi
        self._capi, self._C = _capi, C
        self._lib = _capi.load()
        self.rank, self.world = rank, world
        self._h = C.c_void_p()
        if inproc_group is not None:
            _capi.check(self._lib.coala_comm_create_inproc(inproc_group, rank, int(device_index), C.byref(self._h)))
        else:
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                _capi.check(self._lib.coala_comm_unique_id(uid.data_ptr(), 128))
            if world > 1:
                dist.broadcast(uid, src=src_global_rank, group=bootstrap_group)
            _capi.check(self._lib.coala_comm_create(uid.data_ptr(), rank, world, int(device_index), C.byref(self._h)))
        if rounds is not None:
            _capi.check(self._lib.coala_comm_set_rounds(self._h, int(rounds)))
        self.rccl_ranks = int(self._lib.coala_comm_size(self._h))   # as the transport counts them (ncclCommCount), not the number asked for
        if self.rccl_ranks != world:
            raise RuntimeError(f"the exchange's communicator reports {self.rccl_ranks} ranks, {world} expected: {_capi.last_error()}")
        self._tickets = 0        # count exchanges issued ahead so far (tickets are consecutive)
        self.last_send_counts = self.last_recv_counts = None
        self._profile = False

    @property
    def rounds(self):
        return int(self._lib.coala_comm_get_rounds(self._h))

    @rounds.setter
    def rounds(self, k):  # must be set to the same value on every rank, between fetches
        self._capi.check(self._lib.coala_comm_set_rounds(self._h, int(k)))

    def fetch_events(self, mode=1):
        """coala_comm_fetch_events: 1 = a bucketed fetch hands out its end events (one per stream; nothing of them is a packet on the caller's
        stream), 2 = also a begin event on the probe's launch (for a timer; costs that launch ~5 us), 0 = off."""
        mode = int(mode)
        if mode != getattr(self, "_fetch_events_mode", None):
            self._capi.check(self._lib.coala_comm_fetch_events(self._h, mode))
            self._fetch_events_mode = mode

    def last_fetch_events(self):
        """(begin, end on the caller's stream, end on the communicator's stream) of the most recent fetch as native handles; None where there is none."""
        C = self._C
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._capi.check(self._lib.coala_comm_last_fetch_events(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def set_self_loopback(self, on):
        """Diagnostics (coala_comm_set_self_loopback): the own segment takes the road of a peer's, through the transport itself."""
        self._capi.check(self._lib.coala_comm_set_self_loopback(self._h, 1 if on else 0))

    # same profiling surface as AllToAllExchange (bench.py's xGMI figure)
    @property
    def profile(self):
        return self._profile

    @profile.setter
    def profile(self, on):
        self._profile = bool(on)
        self._capi.check(self._lib.coala_comm_profile(self._h, 1 if on else 0, None, 0))

    def fold_profile(self):
        p = self._capi.CommProfile()
        self._capi.check(self._lib.coala_comm_profile(self._h, -1, self._C.byref(p), 0))
        return p.rows_ms, int(p.calls), int(p.remote_rows_in)

    def reset_profile(self):
        self._capi.check(self._lib.coala_comm_profile(self._h, -1, None, 1))

    def fetch(self, ops, out_ptr, idx_ptr, n, max_index=0):
        from COALA_GNN_Pybind import current_stream, native
        if native is not None:   # compiled binding: no ctypes marshalling on the per-step path
            native.cache_fetch_distributed(ops._h.value or 0, self._h.value or 0, int(out_ptr), int(idx_ptr), int(n), current_stream())
        else:
            self._capi.check(self._lib.coala_cache_fetch_distributed(ops._h, self._h, int(out_ptr) or None, int(idx_ptr) or None, int(n),
                                                                      current_stream()))
        send = (self._C.c_int64 * self.world)()
        recv = (self._C.c_int64 * self.world)()
        self._lib.coala_comm_last_counts(self._h, send, recv)
        self.last_send_counts, self.last_recv_counts = list(send), list(recv)

    def counts_begin(self, counts_ptr):
        """Issue the count exchange of a later fetch_bucketed now, on the current stream (collective; no host wait) -> ticket."""
        from COALA_GNN_Pybind import current_stream
        t = self._C.c_int64(-1)
        self._capi.check(self._lib.coala_comm_counts_begin(self._h, int(counts_ptr), current_stream(), self._C.byref(t)))
        self._tickets = int(t.value) + 1
        return int(t.value)

    def fetch_bucketed(self, ops, out_ptr, idx_ptr, n, counts_ptr, ticket=None):
        """idx already bucketed by owner (NeighborSampler(bucket_by_owner=G)): no routing pass, rows received in place.
        ticket = a counts_begin of the same counts issued earlier: the fetch then runs without a host synchronisation."""
        from COALA_GNN_Pybind import current_stream, native
        if ticket is not None and self._tickets - int(ticket) > self._capi.COUNTS_RING:
            ticket = None   # its slot of the ring has been reused: exchange the counts again, synchronously (the same decision on every rank)
        if ticket is not None:
            self._capi.check(self._lib.coala_cache_fetch_distributed_bucketed_ahead(ops._h, self._h, int(out_ptr) or None, int(idx_ptr) or None,
                                                                                     int(n), int(ticket), current_stream()))
        elif native is not None:
            native.cache_fetch_distributed_bucketed(ops._h.value or 0, self._h.value or 0, int(out_ptr), int(idx_ptr), int(n), int(counts_ptr),
                                                    current_stream())
        else:
            self._capi.check(self._lib.coala_cache_fetch_distributed_bucketed(ops._h, self._h, int(out_ptr) or None, int(idx_ptr) or None, int(n),
                                                                               int(counts_ptr), current_stream()))
        send = (self._C.c_int64 * self.world)()
        recv = (self._C.c_int64 * self.world)()
        self._lib.coala_comm_last_counts(self._h, send, recv)
        self.last_send_counts, self.last_recv_counts = list(send), list(recv)

    # SSD_GNN_NVSHMEM_Cache.send_requests / read_feature keep the reference's two-call sequence: the second call does it all
    def send_requests(self, ops, idx_ptr, n, req_ptr, max_index):
        self._pending = (idx_ptr, n)

    def read_feature(self, ops, out_ptr, req_ptr, max_index):
        idx_ptr, n = self._pending
        self.fetch(ops, out_ptr, idx_ptr, n, max_index)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.coala_comm_destroy(self._h)
            self._h = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def owner_counts_present(mgr, batch):
    """Will fetch_feature take the bucketed road for this batch (the sampler delivered the input nodes bucketed by owner)?"""
    if mgr.exchange is None or not hasattr(mgr.exchange, "fetch_bucketed") or len(batch) < 3 or not batch[2]:
        return False
    oc = getattr(batch[2][0], "owner_counts", None)
    return oc is not None and oc.numel() == mgr.MPI_comm_manager.local_size and batch[2][0].src_nodes is batch[0]


class COALA_GNN_Manager(object):
    def __init__(self, node_distributor, num_ssds, page_size, num_elems, ssd_read_offset, cache_size,  # MB
                 batch_size, fan_out, dim, MPI_comm_manager, device, cache_backend="nvshmem", sim_buf=None,
                 num_rows=None, profile=False, cold_partitioned=False, exchange=None, out_ring=2):
        self.node_distributor = node_distributor
        self.device = device
        self.cache_backend = cache_backend
        self.MPI_comm_manager = MPI_comm_manager
        self.dim = dim
        self.sim_buf = sim_buf
        self.is_simulation = sim_buf is not None
        self.aggregation_timer = 0.0
        # True (the reference's behaviour): fetch_feature returns when the rows are there and the aggregation timer is host
        # wall time.  False: fully stream-ordered, no host wait; the timer is then fed by a pair of HIP events per call.
        self.sync_on_return = True
        self._agg_events = []
        # Stream-ordered mode.  On a stream that carries nothing but fetches, whatever sits between the cold fill of one minibatch and the
        # probe of the next is idle link (profiles/r04_handover.txt, kernel timestamps, per fetch): plain launches hand over without a gap; ONE
        # event recorded behind the fetch costs 5.8 us; begin / end events riding on the two dispatches (coala_cache_fetch_events) 14.4 us --
        # an attached event makes its kernel wait for, and be waited for by, its neighbours; a recorded timing pair + completion event 15.4 us.
        # So: the aggregation timer's event pair goes on every timing_stride-th fetch only and counts timing_stride times (1 = every fetch, the
        # default; COALA_GNN_DataLoader sets 16; 0 = no timing at all: a caller that never reads the timer), and completion is left to the caller's one event (last_done_event None) -- except for the
        # native exchange's bucketed fetch, whose events ride on its own launches (coala_comm_fetch_events: begin on the probe, one end event per
        # stream; measured against recorded ones in profiles/r04_dist_fetch_packets.txt): last_done_event is then a tuple of native handles.
        self.timing_stride = 1
        self._fetch_no = 0
        self.last_done_event = None
        self._native_exchange_events = False
        if not self.is_simulation:
            raise RuntimeError("sim_buf is None: the NVMe/BaM tier is out of scope here; pass the pinned feature table "
                               "(the reference's --feat_cpu mode, used by every published script)")
        if num_rows is None and hasattr(sim_buf, "shape"):
            num_rows = int(sim_buf.shape[0])
        sim_ptr = int(sim_buf.data_ptr())

        # the GPU this rank drives: its local rank, unless the topology object pins another ordinal (several ranks on one GPU)
        device_id = getattr(MPI_comm_manager, "device_index", MPI_comm_manager.local_rank)
        self.SSD_Controllers = SSD_GNN_SSD_Controllers(num_ssds, page_size, num_elems, ssd_read_offset, device_id, dim,
                                                       self.is_simulation)
        self.max_sample_size = batch_size                     # :79-81
        for i in fan_out:
            self.max_sample_size *= (int(i) + 1)

        dm = None if node_distributor is None else node_distributor.distribute_manager
        G = MPI_comm_manager.local_size
        self.exchange = None
        # "native": the fused C call with its own RCCL communicator -- the default whenever the cache group really is an RCCL
        # group of more than one rank.  "torch": the same sequence driven from Python over torch.distributed (the default for
        # one rank, and the only choice over a gloo group).  COALA_EXCHANGE overrides.
        import os
        if exchange is None:
            exchange = os.environ.get("COALA_EXCHANGE")
        if exchange is None:
            grp = MPI_comm_manager.nccl_cache_gather
            rccl = G > 1 and grp is not None and dist.is_initialized() and dist.get_backend(grp) == "nccl"
            exchange = "native" if rccl else "torch"
        if exchange not in ("torch", "native"):
            raise ValueError("exchange must be 'torch' or 'native'")
        self.exchange_kind = exchange

        def make_exchange():
            if exchange == "native":
                return NativeExchange(MPI_comm_manager.local_gloo_gather, MPI_comm_manager.master_process_id,
                                      MPI_comm_manager.local_rank, G, device_id)
            return AllToAllExchange(MPI_comm_manager.nccl_cache_gather, MPI_comm_manager.local_rank, G, dim, self.device)
        if self.cache_backend == "nvshmem":                   # :83-99
            self.nvshmem_manager = NVSHMEM_Manager(0, MPI_comm_manager.local_rank)
            self.NVshmem_tensor_manager = NVShmem_Tensor_Manager(self.max_sample_size, dim, G, self.device)
            self.COALA_GNN_Cache = SSD_GNN_NVSHMEM_Cache(self.SSD_Controllers, dm, MPI_comm_manager.global_rank, G, cache_size,
                                                         sim_ptr, num_rows=num_rows, profile=profile, sync=False,
                                                         max_batch=self.max_sample_size, rank=MPI_comm_manager.local_rank,
                                                         cold_partitioned=cold_partitioned)
            self.exchange = make_exchange()
            self.COALA_GNN_Cache.attach_exchange(self.exchange)
        elif self.cache_backend in ("isolated", "nccl"):      # :101-111
            self.COALA_GNN_Cache = Isolated_Cache(self.SSD_Controllers, dm, MPI_comm_manager.global_rank, G, cache_size,
                                                  sim_ptr, num_rows=num_rows, profile=profile, sync=False,
                                                  max_batch=self.max_sample_size, rank=MPI_comm_manager.local_rank,
                                                  cold_partitioned=cold_partitioned and self.cache_backend == "nccl")
            if cold_partitioned and self.cache_backend == "isolated":
                raise ValueError("an isolated cache reads every row: it needs the whole cold table, not an owner's shard")
            if self.cache_backend == "nccl":
                self.exchange = make_exchange()
        else:
            raise ValueError(f"Unsupported cache backend: {self.cache_backend}")  # the reference prints and returns (:113-115)
        if isinstance(self.exchange, NativeExchange) and not profile:
            self.exchange.fetch_events(1)
            self._native_exchange_events = True

    def fetch_feature(self, batch):  # :118-213
        index = batch[0].to(self.device)
        if index.dtype != torch.int64:
            index = index.to(torch.int64)
        index = index.contiguous()
        index_size = len(index)
        index_ptr = index.data_ptr()
        fetch_start = time.time()
        ev_pair = None
        self.last_done_event = None
        native_ev = (not self.sync_on_return) and index_size > 0 and self._native_exchange_events and owner_counts_present(self, batch)
        self._fetch_no += 1
        sampled = self.timing_stride >= 1 and self._fetch_no % self.timing_stride == 0
        if native_ev:
            self.exchange.fetch_events(2 if sampled else 1)      # the begin event rides on the probe only when the timer wants this fetch
        if not self.sync_on_return and not native_ev and sampled:
            ev_pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev_pair[0].record()

        # the sampler may have delivered the ids already bucketed by owner (NeighborSampler(bucket_by_owner=G)): the native exchange
        # then skips its routing pass and receives the rows in place
        owner_counts = None
        if self.exchange is not None and hasattr(self.exchange, "fetch_bucketed") and len(batch) >= 3 and batch[2]:
            oc = getattr(batch[2][0], "owner_counts", None)
            if oc is not None and oc.numel() == self.MPI_comm_manager.local_size and batch[2][0].src_nodes is batch[0]:
                owner_counts = oc
        if owner_counts is not None:
            return_torch = torch.empty([index_size, self.dim], dtype=torch.float, device=self.device)
            ticket = getattr(batch[2][0], "counts_ticket", None)   # the count exchange was issued ahead (loader, counts_ahead=True)
            if ticket is not None:
                self.exchange.fetch_bucketed(self.COALA_GNN_Cache, return_torch.data_ptr(), index_ptr, index_size, owner_counts.data_ptr(), ticket=ticket)
            else:
                self.exchange.fetch_bucketed(self.COALA_GNN_Cache, return_torch.data_ptr(), index_ptr, index_size, owner_counts.data_ptr())
        elif self.cache_backend == "nvshmem":
            return_torch = self.NVshmem_tensor_manager.get_batch_tensor([index_size, self.dim])
            request_tensor_ptr = self.NVshmem_tensor_manager.get_index_tensor_ptr()
            self.COALA_GNN_Cache.send_requests(index_ptr, index_size, request_tensor_ptr, self.max_sample_size)
            self.COALA_GNN_Cache.read_feature(return_torch.data_ptr(), request_tensor_ptr, self.max_sample_size)
        elif self.cache_backend == "isolated":
            # torch.empty, not torch.zeros (:138): the kernels write every row, so the 151 MB memset per batch is dropped
            return_torch = torch.empty([index_size, self.dim], dtype=torch.float, device=self.device)
            self.COALA_GNN_Cache.read_feature(return_torch.data_ptr(), index_ptr, index_size)
        elif self.cache_backend == "nccl":
            return_torch = torch.empty([index_size, self.dim], dtype=torch.float, device=self.device)
            self.exchange.fetch(self.COALA_GNN_Cache, return_torch.data_ptr(), index_ptr, index_size, self.max_sample_size)
        else:
            raise ValueError("Unsupported cache backend for fetch_feature")
        self._keep_index = index
        # The native calls are enqueued on the current stream without synchronising (handles are created with sync=False:
        # one host wait per minibatch instead of one per kernel group); like the reference, fetch_feature returns only when
        # the rows are there, so the aggregation timer measures the whole fetch (COALA_GNN_Manager.py:122,134).
        if self.sync_on_return:
            torch.cuda.current_stream().synchronize()
            self.aggregation_timer += (time.time() - fetch_start)
        elif native_ev:
            a, b_st, b_cs = self.exchange.last_fetch_events()
            if b_st:
                self.last_done_event = (b_st, b_cs) if b_cs else b_st
                if a:
                    self._agg_events.append((a, b_cs or b_st, max(1, self.timing_stride)))
            if len(self._agg_events) >= 64 and (len(self._agg_events) % 64 == 0 or len(self._agg_events) >= 1024):
                # (the communicator keeps 2048 triples: a host that runs far ahead of the device waits here, at 1024 unread ones)
                self._fold_events(wait=len(self._agg_events) >= 1024)
        elif ev_pair is not None:
            ev_pair[1].record()
            self._agg_events.append((ev_pair[0], ev_pair[1], max(1, self.timing_stride)))
            if len(self._agg_events) >= 64:
                self._fold_events(wait=False)
        return (*batch, return_torch)

    def tune_exchange_rounds(self, batches, candidates=(1, 2, 4), reps=None):
        """Pick the number of row-exchange rounds per fetch by measurement: how far the cold fill (PCIe) and the row exchange (xGMI)
        overlap depends on the miss ratio and on the link rates of the machine at hand.  Collective over the cache group: every
        rank passes equally many batches (each is fetched once -- the delivered rows do not depend on the setting), the slowest
        rank's time per setting decides, every rank ends with the same choice.  -> (chosen, {rounds: ms per fetch})"""
        xch = self.exchange
        G = self.MPI_comm_manager.local_size
        if xch is None or G <= 1 or not hasattr(xch, "rounds"):
            return None, {}
        cands = [int(k) for k in candidates]
        reps = int(reps) if reps else len(batches) // len(cands)
        if reps < 1 or reps * len(cands) > len(batches):
            raise ValueError(f"{len(cands)} settings x {max(reps, 1)} fetches need more than {len(batches)} batches")
        # the settings take turns in short blocks (A B C A B C ...): the cache state and the batch sizes drift along the sequence
        block = max(1, reps // 3)
        keep_sync, self.sync_on_return = self.sync_on_return, False
        grp = self.MPI_comm_manager.local_gloo_gather
        ms = torch.zeros(len(cands), dtype=torch.float64)
        done = [0] * len(cands)
        pos, turn = 0, 0
        # a full garbage collection of the interpreter (tens of ms with torch loaded) inside one block would decide the comparison
        import gc
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            while min(done) < reps:
                j = turn % len(cands)
                turn += 1
                take = min(block, reps - done[j])
                if take <= 0:
                    continue
                xch.rounds = cands[j]
                dist.barrier(group=grp)
                torch.cuda.current_stream().synchronize()
                t0 = time.perf_counter()
                for b in batches[pos: pos + take]:
                    self.fetch_feature(b)
                torch.cuda.current_stream().synchronize()
                ms[j] += (time.perf_counter() - t0) * 1e3
                done[j] += take
                pos += take
            ms /= reps
            dist.all_reduce(ms, op=dist.ReduceOp.MAX, group=grp)
        finally:
            self.sync_on_return = keep_sync
            if gc_was_on:
                gc.enable()
        best = cands[int(torch.argmin(ms))]
        xch.rounds = best
        return best, {k: round(float(t), 4) for k, t in zip(cands, ms)}

    def _fold_events(self, wait):
        """Move finished (start, end) event pairs into the aggregation timer; with wait=True, all of them."""
        from COALA_GNN_Pybind import event_elapsed_ms
        keep = []
        for a, b, w in self._agg_events:
            if isinstance(b, int):           # native handles of a distributed fetch: begin on the probe's launch, end behind its last row round
                ms = None if keep else event_elapsed_ms(a, b, wait=wait)   # (events of one stream complete in order)
                if ms is None:
                    keep.append((a, b, w))
                else:
                    self.aggregation_timer += ms * 1e-3 * w
                continue
            if wait:
                b.synchronize()
            if b.query():
                self.aggregation_timer += a.elapsed_time(b) * 1e-3 * w
            else:
                keep.append((a, b, w))
        self._agg_events = keep

    def get_cache_data(self, ptr, n_entries=None):
        self.COALA_GNN_Cache.get_cache_data(ptr, n_entries)

    def print_stats(self):
        self.COALA_GNN_Cache.print_stats()

    def get_aggregate_time(self):
        self._fold_events(wait=True)
        return self.aggregation_timer

    def __del__(self):  # :226-230
        try:
            if self.cache_backend == "nvshmem":
                self.nvshmem_manager.finalize()
            self.COALA_GNN_Cache.close()
        except Exception:
            pass
