"""ctypes binding of libcoala_hip.so (include/coala_hip.h).  This is the stub a maintainer of the reference would add in
place of COALA_GNN_Modules/COALA_GNN_Pybind.cu:27-79: same class surface on top, C ABI underneath.

The library is NEVER replaced by a CPU fallback: if it is missing or does not load, importing this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COALA_HIP_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libcoala_hip.so"))

OK, EINVAL, EHIP, ENOMEM, EIO, EFORMAT, ERANGE, ECOMM = 0, -1, -2, -3, -4, -5, -6, -7
FLAG_SYNC, FLAG_DISTRIBUTED, FLAG_PROFILE, FLAG_COLD_PARTITIONED, FLAG_TAG64 = 1, 2, 4, 8, 16
WAYS = 32
COUNTS_RING = 8   # COALA_COUNTS_RING: count exchanges issued ahead that a communicator keeps (include/coala_hip.h)


class CacheConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("dim", C.c_int32), ("cache_mb", C.c_uint64), ("n_gpus", C.c_int32), ("rank", C.c_int32),
        ("global_rank", C.c_int32), ("flags", C.c_uint32), ("cold_table", C.c_void_p), ("num_rows", C.c_uint64),
        ("node_color", C.c_void_p), ("num_colors", C.c_int32), ("reserved", C.c_int32), ("max_batch", C.c_uint64),
    ]


class CacheGeometry(C.Structure):
    _fields_ = [("num_sets", C.c_uint64), ("num_ways", C.c_uint32), ("cache_dim", C.c_uint32), ("line_bytes", C.c_uint64),
                ("table_bytes", C.c_uint64), ("tag_set_bytes", C.c_uint32), ("reserved", C.c_uint32)]


class CacheProfile(C.Structure):
    _fields_ = [("gather_ms", C.c_double), ("gather_launches", C.c_uint64), ("gather_rows", C.c_uint64),
                ("gather_hits", C.c_uint64), ("fill_ms", C.c_double), ("fill_launches", C.c_uint64),
                ("fill_rows", C.c_uint64), ("event_overhead_us", C.c_double)]


class RowRedirect(C.Structure):
    _fields_ = [("begin", C.c_int64), ("end", C.c_int64), ("out", C.c_void_p), ("row_map", C.c_void_p)]


class SamplerBucketing(C.Structure):
    _fields_ = [("n_parts", C.c_int32), ("reserved", C.c_int32), ("bucketed_nodes", C.c_void_p), ("counts", C.c_void_p),
                ("dst_in_src", C.c_void_p)]


class CommProfile(C.Structure):
    _fields_ = [("rows_ms", C.c_double), ("calls", C.c_uint64), ("remote_rows_in", C.c_uint64)]


# every exported symbol of include/coala_hip.h: name -> (restype, argtypes)
_VP, _I, _I64, _U64, _SZ = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_size_t
SYMBOLS = {
    "coala_last_error": (C.c_char_p, []),
    "coala_abi_version": (_I, []),
    "coala_cache_dim": (_I, [_I]),
    "coala_cache_num_sets": (_U64, [_U64, _I]),
    "coala_cache_create": (_I, [C.POINTER(CacheConfig), C.POINTER(_VP)]),
    "coala_cache_destroy": (_I, [_VP]),
    "coala_cache_geometry": (_I, [_VP, C.POINTER(CacheGeometry)]),
    "coala_cache_read_feature": (_I, [_VP, _VP, _VP, _I64, _VP]),
    "coala_cache_serve": (_I, [_VP, _VP, _VP, _I64, _VP]),
    "coala_cache_serve_probe": (_I, [_VP, _VP, _VP, _I64, _VP]),
    "coala_cache_serve_fill": (_I, [_VP, _VP, _VP, _I64, _I64, _I64, _VP]),
    "coala_cache_serve_fill_ranges": (_I, [_VP, _VP, _VP, _I64, C.POINTER(_I64), C.POINTER(_I64), _I, _VP]),
    "coala_cache_serve_abort": (_I, [_VP, _VP]),
    "coala_cache_serve_probe_redirect": (_I, [_VP, _VP, _VP, _I64, C.POINTER(RowRedirect), _VP]),
    "coala_cache_route": (_I, [_VP, _VP, _I64, _I, _I64, _VP, _VP, _VP, _VP, _VP]),
    "coala_cache_scatter": (_I, [_VP, _VP, _VP, _VP, _I64, _VP]),
    "coala_cache_scatter_ranges": (_I, [_VP, _VP, _VP, _VP, C.POINTER(_I64), C.POINTER(_I64), _I, _VP]),
    "coala_cache_row_dim": (_I64, [_VP]),
    "coala_cache_fetch_events": (_I, [_VP, _I]),
    "coala_cache_last_fetch_events": (_I, [_VP, C.POINTER(_VP), C.POINTER(_VP)]),
    "coala_stream_wait_event": (_I, [_VP, _VP]),
    "coala_event_elapsed_ms": (_I, [_VP, _VP, _I, C.POINTER(C.c_float)]),
    "coala_comm_unique_id": (_I, [_VP, _SZ]),
    "coala_comm_create": (_I, [_VP, _I, _I, _I, C.POINTER(_VP)]),
    "coala_comm_destroy": (_I, [_VP]),
    "coala_comm_size": (_I, [_VP]),
    "coala_comm_group_create": (_I, [_I, C.POINTER(_VP)]),
    "coala_comm_group_destroy": (_I, [_VP]),
    "coala_comm_create_inproc": (_I, [_VP, _I, _I, C.POINTER(_VP)]),
    "coala_comm_set_rounds": (_I, [_VP, _I]),
    "coala_comm_get_rounds": (_I, [_VP]),
    "coala_comm_set_self_loopback": (_I, [_VP, _I]),
    "coala_comm_fetch_events": (_I, [_VP, _I]),
    "coala_comm_last_fetch_events": (_I, [_VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP)]),
    "coala_comm_profile": (_I, [_VP, _I, C.POINTER(CommProfile), _I]),
    "coala_comm_last_counts": (_I, [_VP, _VP, _VP]),
    "coala_cache_fetch_distributed": (_I, [_VP, _VP, _VP, _VP, _I64, _VP]),
    "coala_cache_fetch_distributed_bucketed": (_I, [_VP, _VP, _VP, _VP, _I64, _VP, _VP]),
    "coala_comm_counts_begin": (_I, [_VP, _VP, _VP, _VP]),
    "coala_cache_fetch_distributed_bucketed_ahead": (_I, [_VP, _VP, _VP, _VP, _I64, _I64, _VP]),
    "coala_cache_color_counts": (_I, [_VP, _VP, C.c_int32, _VP]),
    "coala_cache_color_counts_async": (_I, [_VP, C.c_int32, _VP]),
    "coala_cache_color_counts_finish": (_I, [_VP, _VP, C.c_int32]),
    "coala_cache_stats": (_I, [_VP, C.POINTER(_U64), C.POINTER(_U64), C.POINTER(_U64), _I, _VP]),
    "coala_cache_dump": (_I, [_VP, _VP, _VP, _VP, _VP]),
    "coala_cache_profile": (_I, [_VP, C.POINTER(CacheProfile), _I]),
    "coala_coloring_create": (_I, [_U64, C.POINTER(_VP)]),
    "coala_coloring_destroy": (_I, [_VP]),
    "coala_coloring_set_adj_csc": (_I, [_VP, _VP, _VP]),
    "coala_coloring_set_color_buffer": (_I, [_VP, _VP]),
    "coala_coloring_set_topk_buffers": (_I, [_VP, _VP, _VP, _I]),
    "coala_coloring_color_optimized": (_I, [_VP, _VP, _U64, C.c_uint]),
    "coala_coloring_color_all": (_I, [_VP, C.c_uint]),
    "coala_coloring_num_color": (_U64, [_VP]),
    "coala_coloring_num_color_node": (_U64, [_VP]),
    "coala_coloring_topk": (_I, [_VP, _I]),
    "coala_coloring_nearest": (_I, [_VP]),
    "coala_sampler_create": (_I, [_I, _VP, _VP, _I64, _I64, C.POINTER(_VP)]),
    "coala_sampler_destroy": (_I, [_VP]),
    "coala_sampler_sample": (_I, [_VP, _VP, _I64, C.POINTER(C.c_int32), _I, _U64, _U64, C.POINTER(_VP), C.POINTER(_VP),
                             C.POINTER(_I64), C.POINTER(SamplerBucketing), C.POINTER(_I64), _VP]),
    "coala_block_mean_aggregate": (_I, [_I, _VP, _VP, _VP, _I64, _I, _I, _VP]),
    "coala_block_mean_aggregate_backward": (_I, [_I, _VP, _VP, _VP, _I64, _I, _I, _VP]),
    "coala_sampler_wait": (_I, [_VP, _I64, C.POINTER(_I64), C.POINTER(_I64)]),
    "coala_shm_open": (_I, [C.c_char_p, _U64, _I, _I, C.POINTER(_VP)]),
    "coala_shm_host_ptr": (_VP, [_VP]),
    "coala_shm_device_ptr": (_VP, [_VP]),
    "coala_shm_close": (_I, [_VP, _I]),
    "coala_pinned_alloc": (_I, [_U64, _I, C.POINTER(_VP), C.POINTER(_VP)]),
    "coala_pinned_free": (_I, [_VP]),
    "coala_device_pci_bus_id": (_I, [_I, C.c_char_p, _SZ]),
    "coala_npy_parse": (_I, [C.c_char_p, _SZ, _I, C.POINTER(_I64), C.POINTER(_I), C.POINTER(_SZ), C.c_char_p, _SZ]),
    "coala_distributor_create_plain": (_I, [_VP, _I, C.POINTER(_VP)]),
    "coala_distributor_create": (_I, [_VP, _I, _I, _I, _I, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(_VP)]),
    "coala_distributor_destroy": (_I, [_VP]),
    "coala_distributor_num_colors": (_I, [_VP]),
    "coala_distributor_color_ptr": (_VP, [_VP]),
    "coala_distributor_num_color_entries": (_I64, [_VP]),
    "coala_distributor_assign": (_I, [_VP, _U64, _VP, C.POINTER(_VP), _I]),
}

_lib = None


def load():
    """dlopen the HIP library and type every entry point.  Raises if it is absent: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"libcoala_hip.so not found at {LIB_PATH}: build it with `python coala-gnn_amd/build.py` "
            "(hipcc --offload-arch=gfx950).  The product has no CPU fallback.")
    # One HIP runtime and one RCCL per process: the PyTorch-ROCm wheel bundles its own libamdhip64.so / librccl.so (same
    # SONAMEs as /opt/rocm's).  If torch is imported first, our NEEDED entries resolve to the copies it already loaded; the
    # other order maps BOTH copies (seen as "double free or corruption" at exit).  So: torch first, whenever it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def last_error():
    msg = load().coala_last_error()
    return msg.decode(errors="replace") if msg else ""


def check(rc):
    if rc != OK:
        raise RuntimeError(f"libcoala_hip: {last_error()} (code {rc})")
    return rc
