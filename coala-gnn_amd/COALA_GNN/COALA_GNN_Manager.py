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
import time

import torch
import torch.distributed as dist

from COALA_GNN_Pybind import NVSHMEM_Manager, SSD_GNN_SSD_Controllers, SSD_GNN_NVSHMEM_Cache, Isolated_Cache

__all__ = ["COALA_GNN_Manager", "NVShmem_Tensor_Manager", "AllToAllExchange", "NativeExchange"]


class NVShmem_Tensor_Manager(object):
    """COALA_GNN_Manager.py:8-40.  Owns the per-step output buffer ring and the request buffer (plain HBM)."""

    def __init__(self, max_rows, dim, n_gpus, device, ring=2):
        self.device = device
        self.dim = dim
        self.batch = [torch.empty((max_rows, dim), dtype=torch.float32, device=device) for _ in range(ring)]
        self.index_tensor = torch.empty((n_gpus, max_rows * 2), dtype=torch.int64, device=device)
        self._turn = 0

    def get_batch_tensor(self, shape):
        # ring of buffers: the tensor returned for step t stays valid while step t+1 is being fetched (the reference
        # returns a view of ONE symmetric buffer every step, COALA_GNN_Manager.py:127-128)
        buf = self.batch[self._turn]
        self._turn = (self._turn + 1) % len(self.batch)
        return buf[: shape[0]]

    def get_index_tensor(self):
        return self.index_tensor

    def get_index_tensor_ptr(self):
        return self.index_tensor.data_ptr()


class AllToAllExchange(object):
    """ids out / rows back over one process group.  `ops` provides the device primitives as pointer-level calls
    (route, serve, scatter): a COALA_GNN_Pybind cache object in the product; tests may inject another provider to
    exercise this host logic with gloo on CPU tensors."""

    def __init__(self, group, rank, world, dim, device, stage_through_host=None):
        self.group, self.rank, self.world, self.dim, self.device = group, rank, world, dim, device
        # RCCL moves GPU tensors directly.  A gloo group cannot (no CUDA all-to-all): then the buffers are staged through
        # host memory -- transport only, used by the multi-process tests on a one-GPU box; the kernels stay on the GPU.
        if stage_through_host is None:
            stage_through_host = (world > 1 and str(device).startswith("cuda") and dist.is_initialized()
                                  and dist.get_backend(group) == "gloo")
        self.stage_through_host = bool(stage_through_host)
        self._pending = None
        self.counts = torch.zeros(world, dtype=torch.int64, device=device)
        self.offsets = torch.zeros(world + 1, dtype=torch.int64, device=device)
        self.recv_counts = torch.zeros(world, dtype=torch.int64, device=device)
        on_gpu = str(device).startswith("cuda")
        self._both_dev = torch.zeros((2, world), dtype=torch.int64, device=device)
        # pinned: a D2H copy into pageable memory is staged by the runtime and can wait on more than this stream
        self._both_host = torch.zeros((2, world), dtype=torch.int64, pin_memory=on_gpu)
        self.last_send_counts = None
        self.last_recv_counts = None
        # profile = True brackets the row exchange with HIP events (bench.py's xGMI figure: BASELINE.md "achieved_xGMI")
        self.profile = False
        self._row_events = []
        self.rows_a2a_ms = 0.0
        self.rows_a2a_calls = 0
        self.remote_rows_in = 0

    def fold_profile(self):
        """Finish the pending event pairs; -> (milliseconds in the row all-to-all-v, calls, rows received from other ranks)."""
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

    def send_requests(self, ops, idx_ptr, n, req_ptr, max_index):
        """route + count exchange + id exchange.  ssd_gnn_cache.cuh:111-129 / COALA_GNN_Manager.py:152-165."""
        n = int(n)
        dev = self.device
        node = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
        mp = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
        ops.route(idx_ptr, n, self.world, node.data_ptr(), mp.data_ptr(), self.counts.data_ptr(), self.offsets.data_ptr(), 0)
        self._a2a(self.recv_counts, self.counts)
        self._both_dev[0].copy_(self.counts)
        self._both_dev[1].copy_(self.recv_counts)
        self._both_host.copy_(self._both_dev, non_blocking=True)
        if self._both_dev.is_cuda:
            torch.cuda.current_stream().synchronize()  # the one host synchronisation of the step
        send_c, recv_c = self._both_host[0].tolist(), self._both_host[1].tolist()
        total_recv = int(sum(recv_c))
        recv_ids = torch.empty(max(total_recv, 1), dtype=torch.int64, device=dev)
        self._a2a(recv_ids[:total_recv], node[:n], recv_c, send_c)
        self._pending = (n, node, mp, send_c, recv_c, recv_ids, total_recv)
        self.last_send_counts, self.last_recv_counts = send_c, recv_c

    def read_feature(self, ops, out_ptr, req_ptr, max_index):
        """serve + row exchange + un-permute.  ssd_gnn_cache.cuh:132-174 / COALA_GNN_Manager.py:167-209."""
        n, node, mp, send_c, recv_c, recv_ids, total_recv = self._pending
        self._pending = None
        dev = self.device
        rows_send = torch.empty((max(total_recv, 1), self.dim), dtype=torch.float32, device=dev)
        ops.serve(rows_send.data_ptr(), recv_ids.data_ptr(), total_recv)
        rows_recv = torch.empty((max(n, 1), self.dim), dtype=torch.float32, device=dev)
        timed = self.profile and rows_recv.is_cuda
        if timed:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self._a2a(rows_recv[:n], rows_send[:total_recv], send_c, recv_c)
        if timed:
            ev[1].record()
            self._row_events.append(ev)
            self.rows_a2a_calls += 1
            self.remote_rows_in += n - int(send_c[self.rank])
        ops.scatter(out_ptr, rows_recv.data_ptr(), mp.data_ptr(), n)
        self._keep = (rows_recv, mp)  # until the next step: scatter is stream-ordered, torch's allocator is too

    def fetch(self, ops, out_ptr, idx_ptr, n, max_index=0):
        self.send_requests(ops, idx_ptr, n, 0, max_index)
        self.read_feature(ops, out_ptr, 0, max_index)


class NativeExchange(object):
    """The same exchange as AllToAllExchange, as ONE native call (coala_cache_fetch_distributed: route, ncclAllToAll of the
    counts, ncclAllToAllv of ids and rows, serve, un-permute inside libcoala_hip.so).  Own RCCL communicator per cache
    group, bootstrapped by broadcasting the 128-byte ncclUniqueId over a torch.distributed CPU group."""

    def __init__(self, bootstrap_group, src_global_rank, rank, world, device_index):
        import ctypes as C
        from COALA_GNN_Pybind import _capi
        self._capi, self._C = _capi, C
        self._lib = _capi.load()
        self.rank, self.world = rank, world
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            _capi.check(self._lib.coala_comm_unique_id(uid.data_ptr(), 128))
        if world > 1:
            dist.broadcast(uid, src=src_global_rank, group=bootstrap_group)
        self._h = C.c_void_p()
        _capi.check(self._lib.coala_comm_create(uid.data_ptr(), rank, world, int(device_index), C.byref(self._h)))
        self.last_send_counts = self.last_recv_counts = None

    def fetch(self, ops, out_ptr, idx_ptr, n, max_index=0):
        from COALA_GNN_Pybind import current_stream
        self._capi.check(self._lib.coala_cache_fetch_distributed(ops._h, self._h, int(out_ptr) or None, int(idx_ptr) or None, int(n),
                                                                  current_stream()))
        send = (self._C.c_int64 * self.world)()
        recv = (self._C.c_int64 * self.world)()
        self._lib.coala_comm_last_counts(self._h, send, recv)
        self.last_send_counts, self.last_recv_counts = list(send), list(recv)

    # SSD_GNN_NVSHMEM_Cache.send_requests / read_feature keep the reference's two-call sequence: the first call does it all
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
        # "torch": torch.distributed all_to_all_single on the per-machine RCCL group (default).  "native": the fused C call
        # with its own RCCL communicator (COALA_EXCHANGE=native).
        import os
        exchange = exchange or os.environ.get("COALA_EXCHANGE", "torch")
        if exchange not in ("torch", "native"):
            raise ValueError("exchange must be 'torch' or 'native'")

        def make_exchange():
            if exchange == "native":
                return NativeExchange(MPI_comm_manager.local_gloo_gather, MPI_comm_manager.master_process_id,
                                      MPI_comm_manager.local_rank, G, device_id)
            return AllToAllExchange(MPI_comm_manager.nccl_cache_gather, MPI_comm_manager.local_rank, G, dim, self.device)
        if self.cache_backend == "nvshmem":                   # :83-99
            self.nvshmem_manager = NVSHMEM_Manager(0, MPI_comm_manager.local_rank)
            self.NVshmem_tensor_manager = NVShmem_Tensor_Manager(self.max_sample_size, dim, G, self.device, ring=max(2, int(out_ring)))
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

    def fetch_feature(self, batch):  # :118-213
        index = batch[0].to(self.device)
        if index.dtype != torch.int64:
            index = index.to(torch.int64)
        index = index.contiguous()
        index_size = len(index)
        index_ptr = index.data_ptr()
        fetch_start = time.time()
        ev_pair = None
        if not self.sync_on_return:
            ev_pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev_pair[0].record()

        if self.cache_backend == "nvshmem":
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
        else:
            ev_pair[1].record()
            self._agg_events.append(ev_pair)
            if len(self._agg_events) >= 64:
                self._fold_events(wait=False)
        return (*batch, return_torch)

    def _fold_events(self, wait):
        """Move finished (start, end) event pairs into the aggregation timer; with wait=True, all of them."""
        keep = []
        for a, b in self._agg_events:
            if wait:
                b.synchronize()
            if b.query():
                self.aggregation_timer += a.elapsed_time(b) * 1e-3
            else:
                keep.append((a, b))
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
