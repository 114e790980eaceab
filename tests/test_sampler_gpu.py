"""GPU tests of the native neighbour sampler (coala_sampler.hip) through the C ABI.

The reference's sampler is DGL 2.5 (README.md:12), absent from /root/reference: PARITY UNPINNED.  These tests pin the
build's own contract: (i) the DGL sampler's published properties (each sampled neighbour is an in-neighbour of its dst,
count = min(deg, fanout), distinct positions, input nodes unique with the dst nodes first, same-seed determinism) and
(ii) bit-exact agreement with the CPU twin in oracle/coala_oracle.c (same counter-based RNG)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _graph(torch, n, deg, seed):
    from COALA_GNN.synthetic import powerlaw_csc
    return powerlaw_csc(n, deg, seed=seed, device="cuda")


@pytest.mark.parametrize("n_nodes,avg_deg,fanouts,n_seeds", [
    (2000, 4.0, [5, 5], 64),
    (50000, 12.0, [5, 5], 1024),
    (50000, 12.0, [10, 5, 5], 256),
    (30000, 25.0, [15, 10, 5], 300),
    (1000, 2.0, [3], 1),
])
def test_sampler_matches_cpu_twin_and_properties(hiplib, oracle, n_nodes, avg_deg, fanouts, n_seeds):
    import torch
    from COALA_GNN.sampler import NeighborSampler
    indptr, indices = _graph(torch, n_nodes, avg_deg, seed=3)
    ip, ix = indptr.cpu().numpy(), indices.cpu().numpy()
    sampler = NeighborSampler(fanouts, seed=11)
    g = sampler.make_graph(indptr, indices, ndata={"labels": torch.arange(n_nodes, device="cuda") % 19})
    gen = torch.Generator().manual_seed(1)
    for step in range(3):
        seeds = torch.randperm(n_nodes, generator=gen)[:n_seeds].cuda()
        input_nodes, out_nodes, blocks = sampler.sample(g, seeds)
        assert torch.equal(out_nodes, seeds)
        twin = oracle.sample_blocks(ip, ix, seeds.cpu().numpy(), list(reversed(fanouts)), 11, step)
        assert len(blocks) == len(fanouts)
        dst = seeds.cpu().numpy()
        for l, (src_t, local_t, nbr_t) in enumerate(twin):
            b = blocks[len(fanouts) - 1 - l]  # DGL order: blocks[0] is the input layer
            f = list(reversed(fanouts))[l]
            src = b.src_nodes.cpu().numpy()
            loc = b.nbr.cpu().numpy()
            assert np.array_equal(src, src_t), f"source list differs at layer {l}"
            assert np.array_equal(loc, local_t), f"local indices differ at layer {l}"
            # properties, independent of the twin
            assert np.array_equal(src[: len(dst)], dst)                   # dst nodes first
            assert len(np.unique(src)) == len(src)                        # input nodes unique
            deg = ip[dst + 1] - ip[dst]
            assert np.array_equal((loc >= 0).sum(1), np.minimum(deg, f))  # count = min(deg, fanout)
            for d in range(0, len(dst), max(1, len(dst) // 50)):
                nb = src[loc[d][loc[d] >= 0]]
                col = ix[ip[dst[d]]: ip[dst[d] + 1]]
                assert np.all(np.isin(nb, col))                           # sampled from the CSC column
                if deg[d] > f:                                            # distinct positions -> multiset inclusion
                    u, c = np.unique(nb, return_counts=True)
                    cu, cc = np.unique(col, return_counts=True)
                    assert all(c[i] <= cc[np.searchsorted(cu, u[i])] for i in range(len(u)))
            dst = src
        assert torch.equal(input_nodes, blocks[0].src_nodes)
        assert torch.equal(blocks[-1].dstdata["labels"], seeds % 19)
    # same (seed, step) -> same sample; different step -> different sample
    a = sampler.sample(g, seeds, step=7)[0]
    b = sampler.sample(g, seeds, step=7)[0]
    c = sampler.sample(g, seeds, step=8)[0]
    assert torch.equal(a, b)
    if n_seeds > 32:
        assert not torch.equal(a, c)
    g.close()


def test_sampler_uniformity(hiplib):
    """Every in-neighbour of a high-degree node is picked with probability fanout/deg (chi-square style bound)."""
    import torch
    from COALA_GNN.sampler import NeighborSampler
    deg, f, trials = 40, 5, 4000
    indptr = torch.tensor([0, deg], dtype=torch.int64, device="cuda")
    indptr = torch.cat([indptr, torch.full((deg,), deg, dtype=torch.int64, device="cuda")])  # nodes 1..deg have no in-edges
    indices = torch.arange(1, deg + 1, dtype=torch.int64, device="cuda")
    sampler = NeighborSampler([f], seed=5)
    g = sampler.make_graph(indptr, indices)
    counts = np.zeros(deg + 1)
    seeds = torch.zeros(1, dtype=torch.int64, device="cuda")
    for t in range(trials):
        src = sampler.sample(g, seeds)[0].cpu().numpy()
        assert len(src) == 1 + f
        counts[src[1:]] += 1
    expect = trials * f / deg
    assert np.all(np.abs(counts[1:] - expect) < 6 * np.sqrt(expect))
    g.close()


def test_mean_aggregate_matches_dense(hiplib):
    import torch
    from COALA_GNN.sampler import NeighborSampler
    from COALA_GNN.synthetic import powerlaw_csc
    indptr, indices = powerlaw_csc(5000, 8.0, seed=2, device="cuda")
    sampler = NeighborSampler([4, 4], seed=1)
    g = sampler.make_graph(indptr, indices)
    seeds = torch.arange(0, 5000, 37, device="cuda")
    _, _, blocks = sampler.sample(g, seeds)
    b = blocks[0]
    h = torch.rand(b.num_src, 16, device="cuda")
    got = b.mean_aggregate(h)
    nbr = b.nbr.cpu().numpy()
    hc = h.cpu().numpy()
    want = np.stack([hc[r[r >= 0]].mean(0) if (r >= 0).any() else np.zeros(16, np.float32) for r in nbr])
    assert np.allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-6)  # fp32 mean: 1e-5 relative
    g.close()


@pytest.mark.parametrize("n_nodes,avg_deg,fanouts,n_seeds,G", [(50000, 12.0, [5, 5], 1024, 8), (30000, 12.0, [10, 10], 300, 3),
                                                              (20000, 8.0, [4], 500, 2), (30000, 25.0, [15, 10, 5], 100, 4)])
def test_sampler_owner_bucketing(hiplib, n_nodes, avg_deg, fanouts, n_seeds, G):
    """bucket_by_owner=G: the input nodes come out as a STABLE partition of the unbucketed list by id % G, with the bucket sizes,
    and the input block re-indexed so that it addresses exactly the same nodes -- f-1: the layout the partitioned fetch sends."""
    import torch
    from COALA_GNN.sampler import NeighborSampler
    indptr, indices = _graph(torch, n_nodes, avg_deg, seed=5)
    plain = NeighborSampler(fanouts, seed=3)
    buck = NeighborSampler(fanouts, seed=3, bucket_by_owner=G)
    g = plain.make_graph(indptr, indices, ndata={"labels": torch.arange(n_nodes, device="cuda") % 19})
    gen = torch.Generator().manual_seed(2)
    for step in range(3):
        seeds = torch.randperm(n_nodes, generator=gen)[:n_seeds].cuda()
        in_p, _, bl_p = plain.sample(g, seeds)
        in_b, _, bl_b = buck.sample(g, seeds)
        a, b = bl_p[0], bl_b[0]
        ids = in_p.cpu().numpy()
        want = np.concatenate([ids[ids % G == o] for o in range(G)])            # stable partition
        assert np.array_equal(in_b.cpu().numpy(), want)
        assert b.owner_counts.cpu().tolist() == b.owner_counts_host == [int((ids % G == o).sum()) for o in range(G)]
        assert b.num_src == a.num_src and b.num_dst == a.num_dst
        # same neighbours through the new indices; -1 padding untouched
        na, nb = a.nbr.cpu().numpy(), b.nbr.cpu().numpy()
        assert np.array_equal(na < 0, nb < 0)
        assert np.array_equal(ids[na[na >= 0]], want[nb[nb >= 0]])
        # destination nodes: found through dst_in_src
        dst_nodes = ids[: a.num_dst]
        assert np.array_equal(want[b.dst_in_src.cpu().numpy()], dst_nodes)
        assert torch.equal(b.dstdata["_ID"], a.dstdata["_ID"])
        h = torch.rand(b.num_src, 8, device="cuda")
        assert torch.equal(b.dst_rows(h), h[b.dst_in_src.long()])
        for x, y in zip(bl_p[1:], bl_b[1:]):                                     # the other layers are untouched
            assert torch.equal(x.src_nodes, y.src_nodes) and torch.equal(x.nbr, y.nbr) and y.dst_in_src is None
        if len(fanouts) == 1:
            assert torch.equal(bl_b[-1].dstdata["labels"], seeds % 19)
    g.close()


def test_sampler_async_tickets(hiplib):
    """n_src_host = NULL: the call only enqueues its one kernel; the counts of up to 8 outstanding calls are collected later."""
    import ctypes as C
    import torch
    from COALA_GNN.sampler import NeighborSampler
    from COALA_GNN_Pybind import _capi, current_stream
    L = _capi.load()
    indptr, indices = _graph(torch, 20000, 10.0, seed=1)
    smp = NeighborSampler([5, 5], seed=1)
    g = smp.make_graph(indptr, indices)
    ref = [smp.sample(g, torch.arange(k * 100, k * 100 + 64, device="cuda"), step=k) for k in range(5)]
    tickets, outs = [], []
    for k in range(5):
        seeds = torch.arange(k * 100, k * 100 + 64, device="cuda")
        src = [torch.empty(64 * 6, dtype=torch.int64, device="cuda"), torch.empty(64 * 36, dtype=torch.int64, device="cuda")]
        nbr = [torch.empty(64 * 5, dtype=torch.int32, device="cuda"), torch.empty(64 * 6 * 5, dtype=torch.int32, device="cuda")]
        t = C.c_int64(-1)
        _capi.check(L.coala_sampler_sample(g._h, seeds.data_ptr(), 64, (C.c_int32 * 2)(5, 5), 2, 1, k, (C.c_void_p * 2)(*[x.data_ptr() for x in src]),
                                           (C.c_void_p * 2)(*[x.data_ptr() for x in nbr]), None, None, C.byref(t), current_stream()))
        tickets.append(t.value)
        outs.append((seeds, src, nbr))
    for k in reversed(range(5)):                                                  # any order
        n_src = (C.c_int64 * 2)()
        _capi.check(L.coala_sampler_wait(g._h, tickets[k], n_src, None))
        blocks = ref[k][2]
        assert [n_src[0], n_src[1]] == [blocks[1].num_src, blocks[0].num_src]
        assert torch.equal(outs[k][1][1][: n_src[1]], blocks[0].src_nodes)
    with pytest.raises(RuntimeError, match="ticket"):
        _capi.check(L.coala_sampler_wait(g._h, 99, None, None))
    g.close()


@pytest.mark.parametrize("dim,fan", [(1024, 5), (128, 10), (100, 15), (19, 3)])
def test_native_mean_aggregate_matches_torch(hiplib, dim, fan):
    """Block.mean_aggregate: the native forward / backward kernels against the plain torch fp32 formulation of the same op
    (tolerance 1e-5 relative: the summation order differs; the backward uses hardware float atomics)."""
    import torch
    from COALA_GNN.sampler import NeighborSampler
    from COALA_GNN.synthetic import powerlaw_csc
    indptr, indices = powerlaw_csc(8000, 6.0, seed=2, device="cuda")
    sampler = NeighborSampler([fan, fan], seed=1)
    g = sampler.make_graph(indptr, indices)
    _, _, blocks = sampler.sample(g, torch.arange(0, 8000, 23, device="cuda"))
    for b in blocks:
        h = torch.rand(b.num_src, dim, device="cuda", requires_grad=True)
        h2 = h.detach().clone().requires_grad_(True)
        got = b.mean_aggregate(h)
        want = b.mean_aggregate_torch(h2)
        assert got.shape == want.shape and torch.allclose(got, want, rtol=1e-5, atol=1e-6)
        w = torch.rand_like(got)
        (got * w).sum().backward()
        (want * w).sum().backward()
        assert torch.allclose(h.grad, h2.grad, rtol=1e-5, atol=1e-6)
        # a source tensor that needs no gradient (the input features) costs no backward pass
        feat = torch.rand(b.num_src, dim, device="cuda")
        assert not b.mean_aggregate(feat).requires_grad
    g.close()


def test_sampler_calls_that_change_streams_keep_their_order(hiplib, oracle):
    """The sampler handle's hash table and scan state are ordered by the stream of its calls; calls that alternate between two
    non-blocking streams without any synchronisation in between still equal the CPU twin, call by call."""
    import torch
    from COALA_GNN.sampler import NeighborSampler
    from COALA_GNN.synthetic import powerlaw_csc
    n_nodes = 200_000
    indptr, indices = powerlaw_csc(n_nodes, 10.0, seed=3, device="cuda")
    smp = NeighborSampler([10, 10], seed=5)
    g = smp.make_graph(indptr, indices)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    seeds = [torch.randperm(n_nodes, generator=torch.Generator().manual_seed(s))[:1024].cuda() for s in range(12)]
    torch.cuda.synchronize()
    outs = []
    for lo in (0, 6):  # at most 8 calls may be outstanding: two groups of 6, nothing but the event waits of sample_end in between
        pend = []
        for i in range(lo, lo + 6):
            with torch.cuda.stream(streams[i % 2]):
                pend.append(smp.sample_begin(g, seeds[i], step=i))
        outs += [smp.sample_end(p) for p in pend]
    torch.cuda.synchronize()
    ip, ix = indptr.cpu().numpy(), indices.cpu().numpy()
    for i, (input_nodes, _, blocks) in enumerate(outs):
        want = oracle.sample_blocks(ip, ix, seeds[i].cpu().numpy(), [10, 10], 5, i)
        assert np.array_equal(input_nodes.cpu().numpy(), want[-1][0]), f"call {i}"
    g.close()
