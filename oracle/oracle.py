"""ctypes front-end of the CPU oracle (oracle/coala_oracle.c) plus a tiny pure-numpy mirror.

TEST INFRASTRUCTURE ONLY -- see oracle/coala_oracle.h.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module, and only as the checker.  PARITY UNPINNED BY THE REFERENCE (no fixtures,
not compilable, not importable here); pinned by out == feat[idx] and the committed golden vectors.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcoala_oracle.so")

SCHED_SEQUENTIAL = 0
SCHED_HITS_FIRST = 1
WAYS = 32
EMPTY_KEY = 0xFFFFFFFFFFFFFFFF


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "coala_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "coala_oracle.h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


class _OrcCache(C.Structure):
    _fields_ = [
        ("num_sets", C.c_uint64), ("num_ways", C.c_uint32), ("cache_dim", C.c_uint32), ("dim", C.c_uint32),
        ("keys", C.POINTER(C.c_uint64)), ("set_cnt", C.POINTER(C.c_uint32)), ("color_meta", C.POINTER(C.c_uint64)),
        ("color_counters", C.POINTER(C.c_int32)), ("num_colors", C.c_int32), ("lines", C.POINTER(C.c_float)),
        ("feat", C.c_void_p), ("num_rows", C.c_uint64), ("node_color", C.c_void_p),
        ("hit_cnt", C.c_uint64), ("miss_cnt", C.c_uint64), ("n_gpus", C.c_int32), ("distributed", C.c_int32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_cache_dim.restype = C.c_int
        L.orc_cache_dim.argtypes = [C.c_int]
        L.orc_num_sets.restype = C.c_uint64
        L.orc_num_sets.argtypes = [C.c_uint64, C.c_int]
        L.orc_cache_create.restype = C.POINTER(_OrcCache)
        L.orc_cache_create.argtypes = [C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int,
                                       C.c_int, C.c_int]
        L.orc_cache_destroy.argtypes = [C.POINTER(_OrcCache)]
        L.orc_get_data.restype = C.c_int
        L.orc_get_data.argtypes = [C.POINTER(_OrcCache), C.c_uint64, C.c_void_p]
        L.orc_read_feature.argtypes = [C.POINTER(_OrcCache), C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        L.orc_split_node_list.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64]
        L.orc_map_feat_data.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.orc_dist_fetch.argtypes = [C.POINTER(C.POINTER(_OrcCache)), C.c_int, C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.c_int]
        L.orc_distribute_node_with_affinity.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
                                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                        C.POINTER(C.c_void_p), C.c_void_p]
        L.orc_npy_parse.restype = C.c_int
        L.orc_npy_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int),
                                    C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
        L.orc_feat_value.restype = C.c_float
        L.orc_feat_value.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_fill_features.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_sample_layer.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_uint64,
                                       C.c_uint64, C.c_int, C.c_void_p]
        L.orc_gather_rows_mt.restype = C.c_int
        L.orc_gather_rows_mt.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        L.orc_sample_layer_mt.restype = C.c_int
        L.orc_sample_layer_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_uint64,
                                          C.c_uint64, C.c_int, C.c_void_p, C.c_int]
        L.orc_compact_block.restype = C.c_int64
        L.orc_compact_block.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_compact_block_mt.restype = C.c_int64
        L.orc_compact_block_mt.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data


def cache_dim(dim):
    return lib().orc_cache_dim(int(dim))


def num_sets(cache_mb, cdim):
    return lib().orc_num_sets(int(cache_mb), int(cdim))


def feat_value(row, col, seed):
    return lib().orc_feat_value(int(row), int(col), int(seed))


def make_features(num_rows, dim, seed=0, row0=0):
    """fp32 [num_rows, dim] synthetic table of BASELINE.md section 4 (vectorised numpy; the C twin is orc_fill_features)."""
    r = (np.arange(row0, row0 + num_rows, dtype=np.uint64) * np.uint64(0x9E3779B1)).astype(np.uint32)
    c = (np.arange(dim, dtype=np.uint64) * np.uint64(0x85EBCA77)).astype(np.uint32)
    u = r[:, None] + c[None, :] + np.uint32(seed)
    return ((u >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


class OracleCache:
    """Sequential restatement of Isolated_cache_d_t / NVSHMEM_cache_d_t + their host front-ends."""

    def __init__(self, cache_mb, dim, feat, node_color=None, num_colors=0, n_gpus=1, distributed=False,
                 tag_only=False):
        assert feat.dtype == np.float32 and feat.flags.c_contiguous and feat.shape[1] == dim
        self._feat = feat
        self._color = None
        if node_color is not None:
            self._color = np.ascontiguousarray(node_color, dtype=np.int64)
        self._h = lib().orc_cache_create(int(cache_mb), int(dim), _ptr(feat), feat.shape[0], _ptr(self._color),
                                         int(num_colors), int(n_gpus), int(bool(distributed)), int(bool(tag_only)))
        if not self._h:
            raise RuntimeError("orc_cache_create failed (dim > 1024 or zero sets)")
        self.dim = dim
        self.num_colors = num_colors

    def close(self):
        if self._h:
            lib().orc_cache_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_sets(self):
        return int(self._h.contents.num_sets)

    @property
    def cache_dim(self):
        return int(self._h.contents.cache_dim)

    @property
    def hit_cnt(self):
        return int(self._h.contents.hit_cnt)

    @property
    def miss_cnt(self):
        return int(self._h.contents.miss_cnt)

    def reset_stats(self):
        self._h.contents.hit_cnt = 0
        self._h.contents.miss_cnt = 0

    def keys(self):
        n = self.num_sets * WAYS
        return np.ctypeslib.as_array(self._h.contents.keys, shape=(n,)).reshape(self.num_sets, WAYS).copy()

    def set_cnt(self):
        return np.ctypeslib.as_array(self._h.contents.set_cnt, shape=(self.num_sets,)).copy()

    def color_meta(self):
        n = self.num_sets * WAYS
        return np.ctypeslib.as_array(self._h.contents.color_meta, shape=(n,)).reshape(self.num_sets, WAYS).copy()

    def color_counters(self):
        return np.ctypeslib.as_array(self._h.contents.color_counters, shape=(self.num_colors + 1,)).copy()

    def get_data(self, node_id):
        out = np.empty(self.dim, dtype=np.float32)
        hit = lib().orc_get_data(self._h, int(node_id), _ptr(out))
        return out, bool(hit)

    def read_feature(self, idx, schedule=SCHED_HITS_FIRST, want_rows=True):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        out = np.empty((len(idx), self.dim), dtype=np.float32) if want_rows else None
        lib().orc_read_feature(self._h, _ptr(idx), len(idx), _ptr(out), int(schedule))
        return out


def split_node_list(idx, local_size, max_sample):
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    node = np.zeros(local_size * max_sample, dtype=np.int64)
    mp = np.zeros(local_size * max_sample, dtype=np.int64)
    cnt = np.zeros(local_size, dtype=np.int64)
    lib().orc_split_node_list(_ptr(idx), len(idx), _ptr(node), _ptr(mp), _ptr(cnt), local_size, max_sample)
    return node, mp, cnt


def map_feat_data(out, src, mp):
    src = np.ascontiguousarray(src, dtype=np.float32)
    mp = np.ascontiguousarray(mp, dtype=np.int64)
    lib().orc_map_feat_data(_ptr(out), _ptr(src), _ptr(mp), len(mp), out.shape[1])
    return out


def dist_fetch(caches, idx_list, schedule=SCHED_HITS_FIRST, want_rows=True):
    """One collective step over G logical ranks. Returns list of [n_g, dim] arrays."""
    G = len(caches)
    idx_list = [np.ascontiguousarray(i, dtype=np.int64) for i in idx_list]
    outs = [np.empty((len(i), caches[0].dim), dtype=np.float32) if want_rows else None for i in idx_list]
    cp = (C.POINTER(_OrcCache) * G)(*[c._h for c in caches])
    ip = (C.c_void_p * G)(*[_ptr(i) for i in idx_list])
    np_ = (C.c_int64 * G)(*[len(i) for i in idx_list])
    op = (C.c_void_p * G)(*[_ptr(o) for o in outs])
    lib().orc_dist_fetch(cp, G, ip, np_, op, int(schedule))
    return outs


def distribute_node_with_affinity(items, offset, batch_size, local_size, node_id, num_nodes, color, topk, score, meta):
    """node_distributor_pybind.cuh:150-222. meta: list (one per domain) of int32 arrays indexed by colour."""
    items = np.ascontiguousarray(items, dtype=np.int64)
    color = np.ascontiguousarray(color, dtype=np.int64)
    topk = np.ascontiguousarray(topk, dtype=np.int64)
    score = np.ascontiguousarray(score, dtype=np.float64)
    meta = [np.ascontiguousarray(m, dtype=np.int32) for m in meta]
    domain = batch_size * local_size
    out = np.zeros(domain, dtype=np.int64)
    mp = (C.c_void_p * num_nodes)(*[_ptr(m) for m in meta])
    lib().orc_distribute_node_with_affinity(_ptr(items), int(offset), domain * num_nodes, domain, int(node_id),
                                            int(num_nodes), _ptr(color), _ptr(topk), _ptr(score), topk.shape[1], mp,
                                            _ptr(out))
    return out


def sample_blocks(indptr, indices, seeds, fanouts_reversed, seed, step, threads=1):
    """Multi-layer twin of coala_sampler_sample: returns [(src_nodes int64[n_src], nbr_local int32[n_dst, f]), ...].
    threads > 1: the draws of a layer are spread over that many OpenMP threads (same result: the RNG is counter-based); the
    first-appearance compaction uses CAS inserts + a prefix sum and numbers the nodes exactly as the sequential scan does)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int64)
    dst = np.ascontiguousarray(seeds, dtype=np.int64)
    out = []
    for layer, f in enumerate(fanouts_reversed):
        nbr = np.empty(len(dst) * f, dtype=np.int64)
        if threads > 1:
            lib().orc_sample_layer_mt(_ptr(indptr), _ptr(indices), len(indptr) - 1, _ptr(dst), len(dst), int(f), int(seed), int(step),
                                      layer, _ptr(nbr), int(threads))
        else:
            lib().orc_sample_layer(_ptr(indptr), _ptr(indices), len(indptr) - 1, _ptr(dst), len(dst), int(f), int(seed), int(step),
                                   layer, _ptr(nbr))
        src = np.empty(len(dst) * (f + 1), dtype=np.int64)
        local = np.empty(len(dst) * f, dtype=np.int32)
        if threads > 1:
            n_src = lib().orc_compact_block_mt(_ptr(dst), len(dst), _ptr(nbr), int(f), _ptr(src), _ptr(local), int(threads))
        else:
            n_src = lib().orc_compact_block(_ptr(dst), len(dst), _ptr(nbr), int(f), _ptr(src), _ptr(local))
        src = src[:n_src].copy()
        out.append((src, local.reshape(len(dst), f), nbr.reshape(len(dst), f)))
        dst = src
    return out


def gather_rows_mt(table, idx, out, threads):
    """out[i, :] = table[idx[i], :] on `threads` OpenMP threads (row memcpy); -> threads OpenMP actually used."""
    assert table.dtype == np.float32 and table.flags.c_contiguous and out.dtype == np.float32 and out.flags.c_contiguous
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    assert out.shape[0] >= len(idx) and out.shape[1] == table.shape[1]
    return int(lib().orc_gather_rows_mt(_ptr(table), table.shape[1], _ptr(idx), len(idx), _ptr(out), int(threads)))


def npy_parse(buf, want_dim):
    shape = (C.c_int64 * 2)()
    nd = C.c_int(0)
    off = C.c_size_t(0)
    descr = C.create_string_buffer(16)
    rc = lib().orc_npy_parse(buf, len(buf), want_dim, shape, C.byref(nd), C.byref(off), descr, 16)
    return rc, tuple(shape[i] for i in range(nd.value)), off.value, descr.value.decode()


# ----------------------------------------------------------------------------------------------------------------
# Pure-numpy/Python mirror for SMALL cases: an independent second statement of the same algorithm, used by the CPU
# tests to cross-check the C restatement (isolated_cache.h:335-475 under the two schedules).
# ----------------------------------------------------------------------------------------------------------------
class PyMirrorCache:
    def __init__(self, num_sets, dim, feat, node_color=None, num_colors=0, n_gpus=1, distributed=False):
        self.num_sets, self.dim, self.feat = num_sets, dim, feat
        self.keys = np.full((num_sets, WAYS), EMPTY_KEY, dtype=np.uint64)
        self.set_cnt = np.zeros(num_sets, dtype=np.uint32)
        self.color_meta = np.zeros((num_sets, WAYS), dtype=np.uint64)
        self.color_counters = np.zeros(num_colors + 1, dtype=np.int32)
        self.lines = np.zeros((num_sets, WAYS, dim), dtype=np.float32)
        self.node_color = node_color
        self.n_gpus, self.distributed = n_gpus, distributed
        self.hit = self.miss = 0

    def set_id(self, i):
        return (i // self.n_gpus) % self.num_sets if self.distributed else i % self.num_sets

    def _lookup(self, i):
        s = self.set_id(i)
        w = np.nonzero(self.keys[s] == np.uint64(i))[0]
        return s, (int(w[0]) if len(w) else WAYS)

    def _miss(self, i, s):
        w = int(self.set_cnt[s]) % WAYS
        self.set_cnt[s] += 1
        if self.node_color is not None:
            self.color_counters[int(self.color_meta[s, w])] -= 1
            col = int(self.node_color[i])
            self.color_meta[s, w] = col
            self.color_counters[col] += 1
        self.keys[s, w] = i
        self.lines[s, w] = self.feat[i]
        self.miss += 1
        return self.lines[s, w].copy()

    def read_feature(self, idx, schedule=SCHED_HITS_FIRST):
        out = np.zeros((len(idx), self.dim), dtype=np.float32)
        if schedule == SCHED_SEQUENTIAL:
            for p, i in enumerate(idx):
                s, w = self._lookup(int(i))
                if w < WAYS:
                    out[p] = self.lines[s, w]
                    self.hit += 1
                else:
                    out[p] = self._miss(int(i), s)
            return out
        pend = []
        for p, i in enumerate(idx):
            s, w = self._lookup(int(i))
            if w < WAYS:
                out[p] = self.lines[s, w]
                self.hit += 1
            else:
                pend.append((p, int(i), s))
        for p, i, s in pend:
            out[p] = self._miss(i, s)
        return out
