"""Seeded synthetic inputs of the shapes BASELINE.md section 4 names (datasets are not on the GPU box).

Harness plumbing (torch for device memory), not part of the hot path:
  * feature table  feat[r][c] = float(((r*0x9E3779B1 + c*0x85EBCA77 + seed) mod 2^32) >> 8) * 2^-24   (exact in fp32,
    re-derivable anywhere: feature_rows_torch recomputes any rows on the GPU for bit-exact checks without a second copy)
  * power-law CSC graph, int64 indptr/indices, no self loops (examples/create_csc_graph.py:277-279 builds the graph
    straight from edge_index, no self loops added)
  * block colouring + seeded top-k tables shaped like examples/color_info_gen/generate_color_data.py:39-64
"""
import ctypes as C

import numpy as np
import torch

_MASK32 = 0xFFFFFFFF
_K_ROW = 0x9E3779B1
_K_COL = 0x85EBCA77


def feature_rows_torch(ids, dim, seed, out=None):
    """fp32 [len(ids), dim] rows of the synthetic table, computed on ids.device."""
    ids = ids.to(torch.int64)
    r = (ids * _K_ROW) & _MASK32
    c = (torch.arange(dim, dtype=torch.int64, device=ids.device) * _K_COL) & _MASK32
    u = (r[:, None] + c[None, :] + int(seed)) & _MASK32
    vals = (u >> 8).to(torch.float32) * (1.0 / 16777216.0)
    if out is not None:
        out.copy_(vals)
        return out
    return vals


class PinnedFeatureTable:
    """fp32 [rows, dim] cold tier in pinned host memory mapped into the GPU's address space (zero-copy reads)."""

    def __init__(self, num_rows, dim, device=0):
        from COALA_GNN_Pybind import _capi
        self._capi = _capi
        self.rows, self.dim, self.device = int(num_rows), int(dim), int(device)
        self.nbytes = self.rows * self.dim * 4
        hp, dp = C.c_void_p(), C.c_void_p()
        _capi.check(_capi.load().coala_pinned_alloc(self.nbytes, self.device, C.byref(hp), C.byref(dp)))
        self.host_ptr, self.device_ptr = hp.value, dp.value
        buf = (C.c_float * (self.rows * self.dim)).from_address(self.host_ptr)
        self.array = np.frombuffer(buf, dtype=np.float32).reshape(self.rows, self.dim)
        self.cpu_tensor = torch.from_numpy(self.array)

    def data_ptr(self):  # what COALA_GNN_Manager hands to the cache as sim_buf (COALA_GNN_Manager.py:97,103)
        return self.device_ptr

    def close(self):
        if getattr(self, "host_ptr", 0):
            self.cpu_tensor = None
            self.array = None
            self._capi.load().coala_pinned_free(self.host_ptr)
            self.host_ptr = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fill_table(cpu_tensor, seed, device="cuda", chunk_rows=1 << 18, row0=0):
    """Fill a [rows, dim] CPU (pinned) tensor with the synthetic features, generating chunks on the GPU."""
    rows, dim = cpu_tensor.shape
    for lo in range(0, rows, chunk_rows):
        hi = min(rows, lo + chunk_rows)
        ids = torch.arange(row0 + lo, row0 + hi, dtype=torch.int64, device=device)
        cpu_tensor[lo:hi].copy_(feature_rows_torch(ids, dim, seed))
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()


def fill_table_partition(cpu_tensor, seed, rank, world, device="cuda", chunk_rows=1 << 18):
    """Fill an owner's shard of the table: local row k holds node id k*world + rank (COALA_FLAG_COLD_PARTITIONED)."""
    rows, dim = cpu_tensor.shape
    for lo in range(0, rows, chunk_rows):
        hi = min(rows, lo + chunk_rows)
        ids = torch.arange(lo, hi, dtype=torch.int64, device=device) * world + rank
        cpu_tensor[lo:hi].copy_(feature_rows_torch(ids, dim, seed))
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()


def alloc_pinned_table(num_rows, dim, seed, device=0):
    t = PinnedFeatureTable(num_rows, dim, device)
    fill_table(t.cpu_tensor, seed, device=f"cuda:{device}")
    return t


def powerlaw_csc(num_nodes, avg_degree, seed=0, device="cuda", skew=3.0, max_degree=None):
    """Seeded power-law CSC graph: in-degrees ~ Pareto with the requested mean, sources skewed to popular nodes.
    Returns int64 (indptr[N+1], indices[E]) on `device`."""
    g = torch.Generator(device=device).manual_seed(int(seed))
    n = int(num_nodes)
    alpha = 2.5                                    # Pareto tail exponent of the in-degree
    xmin = avg_degree * (alpha - 2.0) / (alpha - 1.0)
    u = torch.rand(n, generator=g, device=device, dtype=torch.float64).clamp_min_(1e-12)
    deg = (xmin * u.pow(-1.0 / (alpha - 1.0))).floor_().to(torch.int64)
    cap = int(max_degree) if max_degree else max(64, int(20 * avg_degree))
    deg.clamp_(min=1, max=cap)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(deg, 0, out=indptr[1:])
    e = int(indptr[-1].item())
    perm = torch.randperm(n, generator=g, device=device)  # popularity rank -> node id
    src = torch.empty(e, dtype=torch.int64, device=device)
    step = 1 << 26
    for lo in range(0, e, step):
        hi = min(e, lo + step)
        r = torch.rand(hi - lo, generator=g, device=device, dtype=torch.float64)
        src[lo:hi] = perm[(r.pow_(skew) * n).to(torch.int64).clamp_(max=n - 1)]
    dst = torch.repeat_interleave(torch.arange(n, device=device, dtype=torch.int64), deg)
    same = src == dst
    src[same] = (src[same] + 1) % n                 # no self loops
    return indptr, src


def community_csc(num_nodes, avg_degree, community=2048, p_in=0.9, seed=0, device="cuda", max_degree=None):
    """Seeded CSC graph with planted communities (blocks of `community` consecutive ids): an in-neighbour comes from the node's
    own block with probability p_in, from anywhere otherwise; in-degrees Pareto as in powerlaw_csc.  The power-law generator
    above has no locality at all, real graphs (IGB, papers100M) do -- this is the shape on which colour-affinity routing has
    something to find.  Returns int64 (indptr[N+1], indices[E]) on `device`."""
    g = torch.Generator(device=device).manual_seed(int(seed))
    n = int(num_nodes)
    alpha = 2.5
    xmin = avg_degree * (alpha - 2.0) / (alpha - 1.0)
    u = torch.rand(n, generator=g, device=device, dtype=torch.float64).clamp_min_(1e-12)
    deg = (xmin * u.pow(-1.0 / (alpha - 1.0))).floor_().to(torch.int64)
    cap = int(max_degree) if max_degree else max(64, int(20 * avg_degree))
    deg.clamp_(min=1, max=cap)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(deg, 0, out=indptr[1:])
    dst = torch.repeat_interleave(torch.arange(n, device=device, dtype=torch.int64), deg)
    e = dst.numel()
    base = (dst // community) * community
    width = torch.clamp(n - base, max=community)
    local = base + (torch.rand(e, generator=g, device=device, dtype=torch.float64) * width.to(torch.float64)).to(torch.int64)
    far = (torch.rand(e, generator=g, device=device, dtype=torch.float64) * n).to(torch.int64).clamp_(max=n - 1)
    src = torch.where(torch.rand(e, generator=g, device=device) < p_in, local, far)
    same = src == dst
    src[same] = torch.where(src[same] + 1 < n, src[same] + 1, src[same] - 1)   # no self loops
    return indptr, src


def block_colors(num_nodes, nodes_per_color=4096, topk=10, seed=0):
    """color[id] = 1 + id // nodes_per_color, with seeded top-k neighbour colours / affinities (numpy, host)."""
    rng = np.random.default_rng(seed)
    color = (1 + np.arange(num_nodes, dtype=np.int64) // nodes_per_color).astype(np.int64)
    num_colors = int(color.max())
    tk = np.zeros((num_colors, topk), dtype=np.int64)
    sc = np.zeros((num_colors, topk), dtype=np.float64)
    for k in range(topk):  # neighbours of colour c: c itself, then c+-1, c+-2 ... with decaying affinity
        off = (k + 1) // 2 * (1 if k % 2 else -1)
        nb = np.arange(1, num_colors + 1, dtype=np.int64) + off
        ok = (nb >= 1) & (nb <= num_colors)
        tk[:, k] = np.where(ok, nb, 0)
        sc[:, k] = np.where(ok, np.exp(-0.5 * ((k + 1) // 2)) * (0.5 + 0.5 * rng.random(num_colors)), 0.0)
    return color, tk, sc, num_colors
