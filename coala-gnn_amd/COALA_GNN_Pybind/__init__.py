"""COALA_GNN_Pybind -- the class surface of the reference's pybind11 module (COALA_GNN_Modules/COALA_GNN_Pybind.cu:27-79)
re-hosted on the MI355X C ABI (include/coala_hip.h, libcoala_hip.so).

Same class names, constructor argument order and method names; every pointer still crosses as a Python int, exactly as
in the reference.  Differences, all documented in INTEGRATION.md:
  * MPI communicator addresses (SharedUVAManager, NVSHMEM_Manager) are accepted and ignored; roles come from the
    explicit `local_rank` keyword or LOCAL_RANK / SLURM_LOCALID.
  * SSD_GNN_NVSHMEM_Cache has no device-initiated transport on this platform: `send_requests` / `read_feature` keep
    their signatures but move ids/rows through an exchange hook (RCCL all-to-all-v, installed by COALA_GNN_Manager).
  * Kernels run on torch's current HIP stream (the reference uses the legacy default stream) and, like the reference,
    every call returns only after its work has completed.
There is no CPU fallback: importing this module loads libcoala_hip.so or raises.
"""
import ctypes as C
import os
import sys

from . import _capi
from ._capi import CacheConfig, CacheGeometry, CacheProfile, check

_lib = _capi.load()

# The compiled binding (csrc/coala_pybind.cpp -> _coala_pybind*.so, built by build.py): the reference's class surface as a
# pybind11 module over the same C ABI.  When present, the per-step calls below go through it (GIL released, no ctypes
# marshalling) and the compiled classes are reachable as COALA_GNN_Pybind.native.<Class>; when absent (not built), everything
# runs through the ctypes table.  Loaded AFTER _capi.load(): torch first, then one libcoala_hip.so for both.
native = None
if os.path.realpath(_capi.LIB_PATH) == os.path.realpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir, "lib", "libcoala_hip.so")):
    # (the compiled module is linked against THAT library: with COALA_HIP_LIB pointing elsewhere -- the development build --
    # handles created by one library must not be driven by the other)
    try:
        from . import _coala_pybind as native
        if native.abi_version() != _lib.coala_abi_version():
            native = None
    except ImportError:
        native = None

__all__ = [
    "SharedUVAManager", "SSD_GNN_SSD_Controllers", "SSD_GNN_NVSHMEM_Cache", "Isolated_Cache", "Node_distributor_pybind",
    "NVSHMEM_Manager", "Graph_Coloring", "current_stream", "set_stream_provider", "native",
]


# ------------------------------------------------------------------------------------------------------------ streams
def _default_stream_provider():
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        return int(torch.cuda.current_stream().cuda_stream)
    return 0


_stream_provider = _default_stream_provider


def set_stream_provider(fn):
    """fn() -> hipStream_t as int.  Default: torch.cuda.current_stream() when torch has initialised the GPU, else 0."""
    global _stream_provider
    _stream_provider = fn


def current_stream():
    return int(_stream_provider())


def stream_wait_event(event, stream=None):
    """The stream (default: the current one) waits for a native event handle (hipStreamWaitEvent; no host wait)."""
    _capi.check(_lib.coala_stream_wait_event(current_stream() if stream is None else stream, event))


def event_elapsed_ms(begin, end, wait=True):
    """Milliseconds between two native event handles; None while `end` has not completed (wait=False)."""
    ms = C.c_float(0.0)
    rc = _lib.coala_event_elapsed_ms(begin, end, 1 if wait else 0, C.byref(ms))
    if rc == 1:
        return None
    _capi.check(rc)
    return float(ms.value)


def _local_rank_from_env():
    for k in ("LOCAL_RANK", "SLURM_LOCALID"):
        if k in os.environ:
            return int(os.environ[k])
    return 0


# ------------------------------------------------------------------------------------------------------------ shm
class SharedUVAManager:
    """shared_UVA.cuh:26-115.  SharedUVAManager(path, bytes, node, global_comm_ptr, local_comm_ptr).

    local rank 0 creates the POSIX shm object; everybody maps it, pins it (hipHostRegister) and gets the device alias.
    `barrier` (callable) stands in for MPI_Barrier(local_comm) between create and open (shared_UVA.cuh:76,79)."""

    def __init__(self, path, shm_size, node=0, global_comm_ptr=0, local_comm_ptr=0, *, local_rank=None, device=None,
                 barrier=None):
        self.path = str(path)
        self.size = int(shm_size)
        self.node_id = int(node)
        self.local_rank = _local_rank_from_env() if local_rank is None else int(local_rank)
        self.device = self.local_rank if device is None else int(device)
        self._h = C.c_void_p()
        creator = self.local_rank == 0
        if creator:
            check(_lib.coala_shm_open(self.path.encode(), self.size, 1, self.device, C.byref(self._h)))
            if barrier:
                barrier()
        else:
            if barrier:
                barrier()
            check(_lib.coala_shm_open(self.path.encode(), self.size, 0, self.device, C.byref(self._h)))
        self._creator = creator

    def get_host_ptr(self):
        return int(_lib.coala_shm_host_ptr(self._h) or 0)

    def get_device_ptr(self):
        return int(_lib.coala_shm_device_ptr(self._h) or 0)

    def cleanup(self):
        if self._h:
            _lib.coala_shm_close(self._h, 1 if self._creator else 0)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------------------ geometry
class SSD_GNN_SSD_Controllers:
    """ssd_gnn_cache.cuh:10-55.  (n_ctrls, page_size, n_elems, read_off, device, dim, sim)."""

    def __init__(self, num_ctrls, p_size, n_elems, read_off, device_id, feat_dim, sim):
        self.n_ctrls = int(num_ctrls)
        self.num_elements = int(n_elems)
        self.offset = int(read_off)
        self.cudaDevice = int(device_id)
        self.dim = int(feat_dim)
        self.SSD_SIM = bool(sim)
        cd = _lib.coala_cache_dim(self.dim)
        if cd < 0:
            raise RuntimeError(_capi.last_error())  # "Only Feature Embedding Size less than 8KB is supported"
        self.cache_dim = cd
        self.page_size = cd * 4  # ssd_gnn_cache.cuh:47 (the p_size argument is overwritten there too)


# ------------------------------------------------------------------------------------------------------------ distributor
class Node_distributor_pybind:
    """node_distributor_pybind.cuh:112-238.  (items_ptr, n_nodes) or
    (items_ptr, node_id, batch, local_size, n_nodes, color_file, topk_file, score_file)."""

    def __init__(self, i_item_ptr, *args):
        self._h = C.c_void_p()
        if len(args) == 1:
            check(_lib.coala_distributor_create_plain(int(i_item_ptr), int(args[0]), C.byref(self._h)))
        elif len(args) == 7:
            n_id, b_size, local_size, n_nodes, color_file, topk_file, score_file = args
            check(_lib.coala_distributor_create(int(i_item_ptr), int(n_id), int(b_size), int(local_size), int(n_nodes),
                                                str(color_file).encode(), str(topk_file).encode(),
                                                str(score_file).encode(), C.byref(self._h)))
        else:
            raise TypeError("Node_distributor_pybind(items, n_nodes) or (items, node_id, batch, local_size, n_nodes, color, topk, score)")

    def distribute_node_with_affinity(self, offset, i_parsed_ptr, meta_data_list):
        n = len(meta_data_list)
        arr = (C.c_void_p * n)(*[int(p) for p in meta_data_list])
        check(_lib.coala_distributor_assign(self._h, int(offset), int(i_parsed_ptr), arr, n))

    def get_num_colors(self):
        return int(_lib.coala_distributor_num_colors(self._h))

    def get_color_buffer_ptr(self):
        return int(_lib.coala_distributor_color_ptr(self._h) or 0)

    def get_num_color_entries(self):
        return int(_lib.coala_distributor_num_color_entries(self._h))

    def __del__(self):
        try:
            if self._h:
                _lib.coala_distributor_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------------------ caches
class _CacheBase:
    def _create(self, ctrls, node_distributer, g_rank, n_gpus, cache_size, sim_b, *, distributed, rank, num_rows,
                profile=False, sync=True, max_batch=0, cold_partitioned=False, tag64=False):
        if not isinstance(ctrls, SSD_GNN_SSD_Controllers):
            raise TypeError("first argument must be SSD_GNN_SSD_Controllers")
        if int(sim_b) == 0:
            raise RuntimeError("sim_buf is 0: the NVMe/BaM storage tier is out of scope on this platform; pass the "
                               "pinned-host feature table (--feat_cpu) as in every published reference script")
        cfg = CacheConfig()
        cfg.device = ctrls.cudaDevice
        cfg.dim = ctrls.dim
        cfg.cache_mb = int(cache_size)
        cfg.n_gpus = int(n_gpus)
        cfg.rank = int(rank)
        cfg.global_rank = int(g_rank)
        cfg.flags = (_capi.FLAG_SYNC if sync else 0) | (_capi.FLAG_DISTRIBUTED if distributed else 0) | (
            _capi.FLAG_PROFILE if profile else 0) | (_capi.FLAG_COLD_PARTITIONED if cold_partitioned else 0) | (
            _capi.FLAG_TAG64 if tag64 else 0)
        cfg.cold_table = int(sim_b)
        color_ptr, num_colors, entries = 0, 0, 0
        if node_distributer is not None:
            color_ptr = node_distributer.get_color_buffer_ptr()
            num_colors = node_distributer.get_num_colors()
            entries = node_distributer.get_num_color_entries()
        if num_rows is None:
            num_rows = entries
        if not num_rows:
            raise RuntimeError("number of feature rows unknown: pass num_rows= or a distributor built with a colour file")
        cfg.num_rows = int(num_rows)
        if color_ptr and entries and entries < cfg.num_rows:
            raise RuntimeError(f"colour table has {entries} entries but the feature table has {cfg.num_rows} rows")
        cfg.node_color = color_ptr
        cfg.num_colors = num_colors
        cfg.max_batch = int(max_batch)
        self._h = C.c_void_p()
        self.dim = ctrls.dim
        self.num_color = num_colors
        self.global_rank = int(g_rank)
        self.local_rank = int(rank)
        self.num_gpus = int(n_gpus)
        check(_lib.coala_cache_create(C.byref(cfg), C.byref(self._h)))
        self._keepalive = node_distributer

    # ssd_gnn_cache.cuh:176-186 / 270-280.  The reference copies num_color entries; n_entries=num_color+1 also returns the
    # last colour.
    def get_cache_data(self, ret_i_ptr, n_entries=None):
        n = self.num_color if n_entries is None else int(n_entries)
        check(_lib.coala_cache_color_counts(self._h, int(ret_i_ptr), n, current_stream()))

    def get_cache_data_async(self, n_entries=None):
        """Enqueue the snapshot of the colour counters at this point of the current stream; no host wait (see get_cache_data_finish)."""
        n = self.num_color if n_entries is None else int(n_entries)
        check(_lib.coala_cache_color_counts_async(self._h, n, current_stream()))

    def get_cache_data_finish(self, ret_i_ptr, n_entries=None):
        """Wait for the pending snapshot (only for it) and store it at ret_i_ptr; callable from a helper thread."""
        n = self.num_color if n_entries is None else int(n_entries)
        check(_lib.coala_cache_color_counts_finish(self._h, int(ret_i_ptr), n))

    def stats(self, reset=False):
        hit, miss, bad = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(_lib.coala_cache_stats(self._h, C.byref(hit), C.byref(miss), C.byref(bad), int(reset), current_stream()))
        return hit.value, miss.value, bad.value

    def print_stats(self):  # isolated_cache.h:132-141 (prints, then resets)
        hit, miss, bad = self.stats(reset=True)
        print(f"Global Rank: {self.global_rank} Local Rank:{self.local_rank} hit count: {hit} miss count: {miss}")
        ratio = hit / (hit + miss) if hit + miss else float("nan")
        print(f"Global Rank: {self.global_rank} Local Rank:{self.local_rank}  GPU hit ratio: {ratio:f}")
        if bad:
            print(f"Global Rank: {self.global_rank} Local Rank:{self.local_rank} rejected ids: {bad}")

    def geometry(self):
        g = CacheGeometry()
        check(_lib.coala_cache_geometry(self._h, C.byref(g)))
        return g

    def profile(self, reset=False):
        p = CacheProfile()
        check(_lib.coala_cache_profile(self._h, C.byref(p), int(reset)))
        return p

    def fetch_events(self, enable=True):
        """coala_cache_fetch_events: begin / end events attached to the kernels of every read_feature (no packets of their own)."""
        _capi.check(_lib.coala_cache_fetch_events(self._h, 1 if enable else 0))

    def last_fetch_events(self):
        """(begin, end) native event handles of the most recent read_feature, or (None, None): wait with stream_wait_event(), time with
        event_elapsed_ms()."""
        a, b = C.c_void_p(), C.c_void_p()
        _capi.check(_lib.coala_cache_last_fetch_events(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def dump(self):
        """(keys[sets,32] u64, set_cnt[sets] u32, color_meta[sets,32] u32) as numpy arrays (test/debug)."""
        import numpy as np
        g = self.geometry()
        keys = np.empty((g.num_sets, 32), dtype=np.uint64)
        cnt = np.empty(g.num_sets, dtype=np.uint32)
        meta = np.empty((g.num_sets, 32), dtype=np.uint32)
        check(_lib.coala_cache_dump(self._h, keys.ctypes.data, cnt.ctypes.data, meta.ctypes.data, current_stream()))
        return keys, cnt, meta

    def close(self):
        if getattr(self, "_h", None):
            _lib.coala_cache_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # primitives shared by both cache classes -------------------------------------------------------------------
    def _read(self, out_ptr, idx_ptr, n):
        if native is not None:
            native.cache_read_feature(self._h.value or 0, int(out_ptr), int(idx_ptr), int(n), current_stream())
        else:
            check(_lib.coala_cache_read_feature(self._h, int(out_ptr), int(idx_ptr), int(n), current_stream()))

    def _serve(self, out_ptr, ids_ptr, n):
        if native is not None:
            native.cache_serve(self._h.value or 0, int(out_ptr), int(ids_ptr), int(n), current_stream())
        else:
            check(_lib.coala_cache_serve(self._h, int(out_ptr), int(ids_ptr), int(n), current_stream()))

    def serve_probe(self, out_ptr, ids_ptr, n):
        """First phase of a split serve: classify the whole batch, copy the hits (coala_cache_serve_probe)."""
        check(_lib.coala_cache_serve_probe(self._h, int(out_ptr), int(ids_ptr), int(n), current_stream()))

    def serve_fill(self, out_ptr, ids_ptr, n, begin, end):
        """Second phase: cold fill of batch positions [begin, end) (coala_cache_serve_fill)."""
        check(_lib.coala_cache_serve_fill(self._h, int(out_ptr), int(ids_ptr), int(n), int(begin), int(end), current_stream()))

    def serve_probe_redirect(self, out_ptr, ids_ptr, n, begin, end, redirect_out_ptr, row_map_ptr=0):
        """serve_probe with the batch positions [begin, end) delivered to redirect_out[row_map[pos - begin]] (the requester's own
        shard of a distributed fetch goes straight into the caller's tensor: coala_cache_serve_probe_redirect)."""
        rd = _capi.RowRedirect(int(begin), int(end), int(redirect_out_ptr) or None, int(row_map_ptr) or None)
        check(_lib.coala_cache_serve_probe_redirect(self._h, int(out_ptr) or None, int(ids_ptr) or None, int(n), C.byref(rd), current_stream()))

    def serve_fill_ranges(self, out_ptr, ids_ptr, n, ranges):
        """Cold fill of a union of disjoint position ranges [(begin, end), ...] of the open batch (coala_cache_serve_fill_ranges)."""
        k = len(ranges)
        b = (C.c_int64 * max(k, 1))(*[int(r[0]) for r in ranges])
        e = (C.c_int64 * max(k, 1))(*[int(r[1]) for r in ranges])
        check(_lib.coala_cache_serve_fill_ranges(self._h, int(out_ptr) or None, int(ids_ptr) or None, int(n), b, e, k, current_stream()))

    def serve_abort(self):
        check(_lib.coala_cache_serve_abort(self._h, current_stream()))

    def scatter_ranges(self, out_ptr, src_ptr, map_ptr, ranges):
        """out[map[r]] = src[r] for the rows r of the ranges [(begin, end), ...] (coala_cache_scatter_ranges)."""
        k = len(ranges)
        b = (C.c_int64 * max(k, 1))(*[int(r[0]) for r in ranges])
        e = (C.c_int64 * max(k, 1))(*[int(r[1]) for r in ranges])
        check(_lib.coala_cache_scatter_ranges(self._h, int(out_ptr), int(src_ptr), int(map_ptr), b, e, k, current_stream()))

    def route(self, idx_ptr, n, n_parts, node_ptr, map_ptr, counts_ptr, offsets_ptr=0, bucket_stride=0):
        check(_lib.coala_cache_route(self._h, int(idx_ptr), int(n), int(n_parts), int(bucket_stride), int(node_ptr),
                                     int(map_ptr), int(counts_ptr), int(offsets_ptr) or None, current_stream()))

    def serve(self, out_ptr, ids_ptr, n):
        self._serve(out_ptr, ids_ptr, n)

    def scatter(self, out_ptr, src_ptr, map_ptr, n):
        check(_lib.coala_cache_scatter(self._h, int(out_ptr), int(src_ptr), int(map_ptr), int(n), current_stream()))


class Isolated_Cache(_CacheBase):
    """ssd_gnn_cache.cuh:201-371.  Isolated_Cache(ctrls, node_distributor, g_rank, n_gpus, cache_MB, sim_buf_ptr).

    Keyword-only extras: num_rows (rows of the feature table when the distributor has no colour file), profile, sync,
    tag64 (keep the reference's 64-bit tags; default: 32-bit tags whenever num_rows < 2^32 -- one 128-B line per set)."""

    def __init__(self, SSD_Controllers, node_distributer, g_rank, n_gpus, cache_size, sim_b, *, num_rows=None,
                 profile=False, sync=True, max_batch=0, rank=None, cold_partitioned=False, tag64=False):
        self._create(SSD_Controllers, node_distributer, g_rank, n_gpus, cache_size, sim_b, distributed=False,
                     rank=SSD_Controllers.cudaDevice % max(int(n_gpus), 1) if rank is None else rank, num_rows=num_rows,
                     profile=profile, sync=sync, max_batch=max_batch, cold_partitioned=cold_partitioned, tag64=tag64)

    def read_feature(self, i_return_tensor_ptr, i_index_ptr, max_index):  # ssd_gnn_cache.cuh:255-268
        self._read(i_return_tensor_ptr, i_index_ptr, max_index)

    def split_node_list(self, i_index_ptr, index_size, i_node_tensor, i_map_tensor, i_counter_tensor, local_size,
                        max_sample):  # ssd_gnn_cache.cuh:283-295 ([G][max_sample] layout)
        self.route(i_index_ptr, index_size, local_size, i_node_tensor, i_map_tensor, i_counter_tensor, 0, max_sample)

    def nccl_get_feature(self, i_index_ptr_list, i_return_ptr_list, index_size_list, local_size, max_sample):
        # ssd_gnn_cache.cuh:297-325.  One batch when the per-peer buffers are contiguous (the contract of DESIGN.md:
        # the owner serves the concatenation in source-rank order); otherwise one batch per peer, in peer order.
        sizes = [int(s) for s in index_size_list][:local_size]
        idxp = [int(p) for p in i_index_ptr_list][:local_size]
        retp = [int(p) for p in i_return_ptr_list][:local_size]
        contiguous = all(idxp[i + 1] == idxp[i] + sizes[i] * 8 and retp[i + 1] == retp[i] + sizes[i] * self.dim * 4
                         for i in range(len(sizes) - 1))
        if contiguous:
            self._serve(retp[0], idxp[0], sum(sizes))
        else:
            for p_idx, p_ret, n in zip(idxp, retp, sizes):
                self._serve(p_ret, p_idx, n)

    def map_feat_data(self, i_return_ptr, i_return_ptr_list, i_meta_buffer, index_size_list, local_size, max_sample):
        # ssd_gnn_cache.cuh:327-356
        for i in range(local_size):
            self.scatter(i_return_ptr, int(i_return_ptr_list[i]), int(i_meta_buffer) + i * int(max_sample) * 8,
                         int(index_size_list[i]))


class SSD_GNN_NVSHMEM_Cache(_CacheBase):
    """ssd_gnn_cache.cuh:58-196.  Same constructor as Isolated_Cache; the table always uses the distributed set index
    (nvshmem_cache.h:347).  The NVSHMEM one-sided transport is replaced by an exchange hook: see attach_exchange()."""

    def __init__(self, SSD_Controllers, node_distributer, g_rank, n_gpus, cache_size, sim_b, *, num_rows=None,
                 profile=False, sync=True, max_batch=0, rank=None, cold_partitioned=False, tag64=False):
        self._create(SSD_Controllers, node_distributer, g_rank, n_gpus, cache_size, sim_b, distributed=True,
                     rank=SSD_Controllers.cudaDevice % max(int(n_gpus), 1) if rank is None else rank, num_rows=num_rows,
                     profile=profile, sync=sync, max_batch=max_batch, cold_partitioned=cold_partitioned, tag64=tag64)
        self._exchange = None

    def attach_exchange(self, exchange):
        """exchange: object with send_requests(cache, idx_ptr, n, req_ptr, max_index) and
        read_feature(cache, out_ptr, req_ptr, max_index) -- COALA_GNN_Manager installs the RCCL all-to-all-v one."""
        self._exchange = exchange

    def send_requests(self, i_src_index_ptr, num_index, i_nvshmem_request_ptr, max_index):  # ssd_gnn_cache.cuh:111-129
        if self._exchange is None:
            raise RuntimeError("SSD_GNN_NVSHMEM_Cache.send_requests: no exchange attached (device-initiated NVSHMEM puts "
                               "do not exist on this platform; COALA_GNN_Manager installs the RCCL all-to-all-v exchange)")
        self._exchange.send_requests(self, i_src_index_ptr, num_index, i_nvshmem_request_ptr, max_index)

    def read_feature(self, i_return_tensor_ptr, i_nvshmem_index_ptr, max_index):  # ssd_gnn_cache.cuh:132-174
        if self._exchange is None:
            raise RuntimeError("SSD_GNN_NVSHMEM_Cache.read_feature: no exchange attached")
        self._exchange.read_feature(self, i_return_tensor_ptr, i_nvshmem_index_ptr, max_index)


# ------------------------------------------------------------------------------------------------------------ symmetric heap
class NVSHMEM_Manager:
    """nvshmem_manager.cuh:9-54.  On this platform the "symmetric heap" is ordinary HBM owned by torch's allocator:
    RCCL collectives need no registered symmetric memory.  allocate(size) -> device pointer (int)."""

    def __init__(self, local_comm_ptr=0, local_rank=None):
        self.local_rank = _local_rank_from_env() if local_rank is None else int(local_rank)
        self._bufs = {}

    def allocate(self, size):
        import torch
        size = int(size)  # 64-bit: the reference's `int size` (nvshmem_manager.cuh:30) overflows past 2 GiB
        if size <= 0:
            raise RuntimeError("NVSHMEM_Manager.allocate: size must be positive")
        t = torch.empty(size, dtype=torch.uint8, device=f"cuda:{self.local_rank}")
        self._bufs[t.data_ptr()] = t
        return t.data_ptr()

    def tensor(self, ptr):
        return self._bufs[int(ptr)]

    def free(self, dest_ptr):
        self._bufs.pop(int(dest_ptr), None)

    def finalize(self):
        self._bufs.clear()


# ------------------------------------------------------------------------------------------------------------ colouring
class Graph_Coloring:
    """graph_coloring.h:15-68 / COALA_GNN_Pybind.cu:65-77.  Graph_Coloring(num_nodes); buffers are host int64 / float64
    tensors passed by address, exactly as examples/color_info_gen/generate_color_data.py:20-54 does."""

    def __init__(self, num_nodes, topk=10, seed=1):
        self._h = C.c_void_p()
        check(_lib.coala_coloring_create(int(num_nodes), C.byref(self._h)))
        self.topk = int(topk)
        self.seed = int(seed)  # 1 == the reference's unseeded glibc rand() stream

    def set_adj_csc(self, i_indp, i_indices):
        check(_lib.coala_coloring_set_adj_csc(self._h, int(i_indp), int(i_indices)))

    def set_color_buffer(self, i_ptr):
        check(_lib.coala_coloring_set_color_buffer(self._h, int(i_ptr)))

    def set_topk_color_buffer(self, i_ptr):
        check(_lib.coala_coloring_set_topk_buffers(self._h, int(i_ptr), None, self.topk))

    def set_topk_affinity_buffer(self, i_ptr):
        check(_lib.coala_coloring_set_topk_buffers(self._h, None, int(i_ptr), self.topk))

    def cpu_color_graph(self):
        check(_lib.coala_coloring_color_all(self._h, self.seed))

    def cpu_color_graph_optimized(self, i_train_node_ptr, num_training_nodes):
        check(_lib.coala_coloring_color_optimized(self._h, int(i_train_node_ptr), int(num_training_nodes), self.seed))

    def cpu_count_nearest_color(self):
        check(_lib.coala_coloring_nearest(self._h))

    def cpu_count_nearest_color_less_memory(self):
        check(_lib.coala_coloring_topk(self._h, 0))

    def cpu_calculate_color_affinity(self):
        check(_lib.coala_coloring_topk(self._h, 1))

    def get_num_color(self):
        return int(_lib.coala_coloring_num_color(self._h))

    def get_num_color_node(self):
        return int(_lib.coala_coloring_num_color_node(self._h))

    def __del__(self):
        try:
            if self._h:
                _lib.coala_coloring_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass
