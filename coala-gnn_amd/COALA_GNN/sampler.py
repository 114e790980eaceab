"""Native neighbour sampler for the COALA_GNN_DataLoader: `NeighborSampler(fanouts).sample(graph, seeds)`.

Stands where the reference passes dgl.dataloading.MultiLayerNeighborSampler (examples/sbatch_ssd_gnn_train.py:70-72; used
at COALA-GNN-Setup/COALA_GNN/COALA_GNN_DataLoader.py:162).  DGL is not installed on the MI355X image; any object with the
same .sample(graph, seeds) -> (input_nodes, output_nodes, blocks) still works with the loader.  The kernels are in
coala-gnn_amd/csrc/coala_sampler.hip (C ABI: coala_sampler_*).  The CSC arrays live in HBM (the reference keeps them in
shared pinned host memory and samples over PCIe: examples/ssd_gnn_dataloader.py:496-523)."""
import ctypes as C

import torch

from COALA_GNN_Pybind import _capi, current_stream

__all__ = ["NeighborSampler", "CSCGraph", "Block"]

_lib = _capi.load()


class CSCGraph(object):
    """int64 CSC (indptr[N+1], indices[E]) resident on one GPU + per-node data (labels...)."""

    def __init__(self, indptr, indices, ndata=None):
        assert indptr.dtype == torch.int64 and indices.dtype == torch.int64
        assert indptr.is_cuda and indices.is_cuda, "the CSC arrays must be device tensors (HBM or a pinned-host alias)"
        self.indptr, self.indices = indptr.contiguous(), indices.contiguous()
        self.num_nodes = self.indptr.numel() - 1
        self.num_edges = self.indices.numel()
        self.device = self.indptr.device
        self.ndata = dict(ndata or {})
        self._h = C.c_void_p()
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _capi.check(_lib.coala_sampler_create(dev, self.indptr.data_ptr(), self.indices.data_ptr(), self.num_nodes,
                                              self.num_edges, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            _lib.coala_sampler_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _MeanAggregate(torch.autograd.Function):
    """out[d] = mean of h_src[nbr[d, j]] over the valid j (coala_block_mean_aggregate): one kernel forward, one backward."""

    @staticmethod
    def forward(ctx, h_src, nbr):
        h = h_src.contiguous()
        n_dst, fanout = nbr.shape
        out = torch.empty((n_dst, h.shape[1]), dtype=torch.float32, device=h.device)
        _capi.check(_lib.coala_block_mean_aggregate(h.device.index or 0, nbr.data_ptr(), h.data_ptr(), out.data_ptr(), n_dst, fanout, h.shape[1],
                                                    current_stream()))
        ctx.save_for_backward(nbr)
        ctx.src_shape = h.shape
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if not ctx.needs_input_grad[0]:
            return None, None
        (nbr,) = ctx.saved_tensors
        g = grad_out.contiguous()
        grad_src = torch.zeros(ctx.src_shape, dtype=torch.float32, device=g.device)
        _capi.check(_lib.coala_block_mean_aggregate_backward(g.device.index or 0, nbr.data_ptr(), g.data_ptr(), grad_src.data_ptr(), nbr.shape[0],
                                                             nbr.shape[1], g.shape[1], current_stream()))
        return grad_src, None


class Block(object):
    """One message-flow block in fixed-stride form: dst node d aggregates src rows nbr[d, j] >= 0.
    The first num_dst source nodes ARE the destination nodes (DGL's to_block convention)."""

    def __init__(self, src_nodes, nbr, num_dst, graph=None, dst_in_src=None, dst_nodes=None, owner_counts=None, owner_counts_host=None):
        self.src_nodes = src_nodes          # int64 [num_src] global ids
        self.nbr = nbr                      # int32 [num_dst, fanout], -1 padded
        self.num_src = int(src_nodes.numel())
        self.num_dst = int(num_dst)
        # Owner-bucketed input layer (NeighborSampler(bucket_by_owner=G)): src_nodes is bucket 0 | bucket 1 | ... (stable inside a
        # bucket) instead of "dst nodes first", and dst_in_src[d] is where the d-th destination node sits in it.
        self.dst_in_src = dst_in_src        # int32 [num_dst] or None
        self.owner_counts = owner_counts    # device int64 [G] or None: what the partitioned fetch sends per owner
        self.owner_counts_host = owner_counts_host
        self.srcdata = {"_ID": src_nodes}
        self.dstdata = {"_ID": src_nodes[: self.num_dst] if dst_nodes is None else dst_nodes}
        if graph is not None:
            for k, v in graph.ndata.items():  # blocks[-1].dstdata['labels'] (examples/sbatch_ssd_gnn_train.py:138)
                self.dstdata[k] = v[self.dstdata["_ID"]] if v.device == src_nodes.device else v[self.dstdata["_ID"].cpu()]

    def dst_rows(self, h_src):
        """Rows of the destination nodes inside a per-source tensor: h_src[:num_dst] (DGL's convention), or a gather through
        dst_in_src when the source list is owner-bucketed."""
        if self.dst_in_src is None:
            return h_src[: self.num_dst]
        return h_src[self.dst_in_src.to(torch.int64)]

    def tensors(self):
        """Every device tensor this block holds (for cross-stream lifetime bookkeeping by the prefetching loader)."""
        yield self.src_nodes
        yield self.nbr
        for t in (self.dst_in_src, self.owner_counts):
            if t is not None:
                yield t
        for d in (self.srcdata, self.dstdata):
            for v in d.values():
                if isinstance(v, torch.Tensor):
                    yield v

    def number_of_src_nodes(self):
        return self.num_src

    def number_of_dst_nodes(self):
        return self.num_dst

    def int(self):   # examples/sbatch_ssd_gnn_train.py:139  block.int().to(device)
        return self

    def to(self, device):
        return self

    def mean_aggregate(self, h_src):
        """Mean of the sampled neighbours' rows for every dst node: fp32 [num_dst, dim] (GraphSAGE 'mean').  Native kernel for
        fp32 rows on the GPU (fan-out <= 32); plain torch otherwise."""
        if h_src.is_cuda and h_src.dtype == torch.float32 and self.nbr.is_cuda and self.nbr.is_contiguous() and self.nbr.shape[1] <= 32:
            return _MeanAggregate.apply(h_src, self.nbr)
        return self.mean_aggregate_torch(h_src)

    def mean_aggregate_torch(self, h_src):
        valid = self.nbr >= 0
        idx = self.nbr.clamp_min(0).to(torch.int64)
        g = h_src[idx] * valid.unsqueeze(-1).to(h_src.dtype)
        return g.sum(1) / valid.sum(1).clamp_min(1).unsqueeze(-1).to(h_src.dtype)


class NeighborSampler(object):
    stream_safe = True  # every kernel and allocation of sample() goes to torch's current stream
    completes_on_host = True  # sample() / sample_end() return after the host has seen the event behind the sample's last kernel (coala_sampler_wait)

    def __init__(self, fanouts, seed=0, bucket_by_owner=0):
        self.fanouts = [int(f) for f in fanouts]
        if not 1 <= len(self.fanouts) <= 8:
            raise ValueError("1..8 layers")
        self.seed = int(seed)
        self.step = 0
        # G > 0: deliver the input nodes bucketed by owner = id % G, the layout the owner-partitioned cache fetches without a
        # routing pass and without an un-permute (blocks[0] then carries dst_in_src / owner_counts; see Block)
        self.bucket_by_owner = int(bucket_by_owner)
        if not 0 <= self.bucket_by_owner <= 64:
            raise ValueError("bucket_by_owner must be 0..64")

    @staticmethod
    def make_graph(indptr, indices, ndata=None):
        return CSCGraph(indptr, indices, ndata)

    def sample(self, g, seed_nodes, step=None):
        """-> (input_nodes, output_nodes, blocks), blocks[0] is the input layer (DGL order)."""
        return self.sample_end(self.sample_begin(g, seed_nodes, step))

    def sample_begin(self, g, seed_nodes, step=None):
        """Enqueue the sample on the current stream and return without waiting (the kernels read their sizes from the device);
        sample_end(pending) collects the counts and builds the blocks.  A caller with something else to enqueue in between -- the
        loader launches step t+1's sample right behind step t's fetch -- never waits for the sampler at all."""
        if isinstance(g, tuple):
            g = CSCGraph(*g)
        seeds = seed_nodes.to(g.device, dtype=torch.int64).contiguous()
        n = seeds.numel()
        rev = list(reversed(self.fanouts))          # DGL samples the output layer first
        L = len(rev)
        caps = [n]
        for f in rev:
            caps.append(caps[-1] * (f + 1))
        src = [torch.empty(max(caps[l + 1], 1), dtype=torch.int64, device=g.device) for l in range(L)]
        nbr = [torch.empty(max(caps[l] * rev[l], 1), dtype=torch.int32, device=g.device) for l in range(L)]
        src_p = (C.c_void_p * L)(*[t.data_ptr() for t in src])
        nbr_p = (C.c_void_p * L)(*[t.data_ptr() for t in nbr])
        fan = (C.c_int32 * L)(*rev)
        st = self.step if step is None else int(step)
        G = self.bucket_by_owner
        bk = None
        extra = None
        if G > 0:
            bucketed = torch.empty(max(caps[L], 1), dtype=torch.int64, device=g.device)
            counts = torch.empty(G, dtype=torch.int64, device=g.device)
            dst_in_src = torch.empty(max(caps[L - 1], 1), dtype=torch.int32, device=g.device)
            bk = _capi.SamplerBucketing(G, 0, bucketed.data_ptr(), counts.data_ptr(), dst_in_src.data_ptr())
            extra = (bucketed, counts, dst_in_src)
        ticket = C.c_int64(-1)
        # three launches per layer, nothing else: no host wait here (n_src_host = NULL)
        _capi.check(_lib.coala_sampler_sample(g._h, seeds.data_ptr(), n, fan, L, self.seed, st, src_p, nbr_p, None,
                                              C.byref(bk) if bk is not None else None, C.byref(ticket), current_stream()))
        if step is None:
            self.step += 1
        return (g, seeds, n, rev, src, nbr, extra, ticket.value)

    def sample_end(self, pending):
        """Wait for the counts of a sample_begin (an event wait: only for that call's kernels) and build the blocks."""
        g, seeds, n, rev, src, nbr, extra, ticket = pending
        L, G = len(rev), self.bucket_by_owner
        n_src = (C.c_int64 * L)()
        ch = (C.c_int64 * G)() if G > 0 else None
        _capi.check(_lib.coala_sampler_wait(g._h, ticket, n_src, ch))
        counts_host = list(ch) if G > 0 else None
        if G > 0:
            bucketed, counts, dst_in_src = extra
        blocks = []
        n_dst = n
        for l in range(L):
            ns = int(n_src[l])
            nbr_l = nbr[l][: n_dst * rev[l]].view(n_dst, rev[l])
            if G > 0 and l == L - 1:   # the input layer: owner-bucketed source list
                blocks.insert(0, Block(bucketed[:ns], nbr_l, n_dst, graph=g if l == 0 else None, dst_in_src=dst_in_src[:n_dst],
                                       dst_nodes=src[l][:n_dst], owner_counts=counts, owner_counts_host=counts_host))
            else:
                blocks.insert(0, Block(src[l][:ns], nbr_l, n_dst, graph=g if l == 0 else None))
            n_dst = ns
        input_nodes = blocks[0].src_nodes
        return input_nodes, seeds, blocks
