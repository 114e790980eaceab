"""GPU parity tests of the isolated cache path, through the C ABI (COALA_GNN_Pybind -> libcoala_hip.so), against the CPU
oracle on the same seeded inputs.  Bit-exact: fp32 row bytes, integer index order, hit/miss/colour counters, tag table.

Reference behaviour under test: Isolated_Cache::read_feature (ssd_gnn_cache.cuh:255-268), get_data
(isolated_cache.h:335-475), split_node_list (ssd_gnn_cache.cuh:283-295), map_feat_data (ssd_gnn_cache.cuh:327-356)."""
import numpy as np
import pytest

from _util import ColorFiles, PinnedTable, synth_colors

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    return torch


def _make_cache(P, table, cache_mb, dist_files=None, items=None, n_gpus=1, cls=None, **kw):
    ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, table.dim, True)
    nd = None
    if dist_files is not None:
        nd = P.Node_distributor_pybind(items.ctypes.data, 0, 4, 1, 1, dist_files.color_file, dist_files.topk_file,
                                       dist_files.score_file)
    cls = cls or P.Isolated_Cache
    return cls(ctrl, nd, 0, n_gpus, cache_mb, table.device_ptr, num_rows=table.rows, **kw), nd


def _compare_tables(cache, orc):
    keys, cnt, meta = cache.dump()
    assert np.array_equal(keys, orc.keys())
    assert np.array_equal(cnt, orc.set_cnt())
    assert np.array_equal(meta.astype(np.uint64), orc.color_meta())


@pytest.mark.parametrize("dim,cache_mb,num_rows", [
    (1024, 1, 6000),   # IGB shape: 4 KiB lines, 8 sets
    (1024, 8, 9000),   # 64 sets
    (128, 1, 20000),   # papers100M shape: 512 B lines, two rows per wave pass
    (100, 1, 20000),   # products shape: dim < cache_dim (SURVEY 3.3 stride fix)
    (512, 2, 5000),
    (256, 1, 5000),
    (200, 1, 5000),    # 256-float lines, partially used
    (99, 1, 4000),     # dim % 4 != 0 -> scalar fallback path
    (1, 1, 3000),
])
@pytest.mark.parametrize("tag64", [False, True], ids=["tags32", "tags64"])   # one 128-B line per set (default below 2^32 rows) / the reference's 64-bit tags
def test_read_feature_matches_oracle(hiplib, oracle, torch_cuda, tmp_path, dim, cache_mb, num_rows, tag64):
    torch = torch_cuda
    P = hiplib
    feat = oracle.make_features(num_rows, dim, seed=3)
    color, tk, sc = synth_colors(num_rows, 12, seed=1)
    files = ColorFiles(tmp_path, color, tk, sc)
    items = np.arange(8, dtype=np.int64)
    table = PinnedTable(P, feat)
    cache, nd = _make_cache(P, table, cache_mb, files, items, tag64=tag64)
    orc = oracle.OracleCache(cache_mb, dim, feat, node_color=color, num_colors=12)
    g = cache.geometry()
    assert (g.num_sets, g.cache_dim, g.tag_set_bytes) == (orc.num_sets, orc.cache_dim, 256 if tag64 else 128)
    rng = np.random.default_rng(7)
    sizes = [1, 3, 63, 64, 65, 257, 1000, min(4097, num_rows), min(2500, num_rows), 5]
    hot = rng.choice(num_rows, size=min(num_rows, 700), replace=False)  # a working set that produces hits
    for n in sizes:
        if rng.random() < 0.5:
            idx = rng.choice(num_rows, size=n, replace=False)
        else:
            idx = rng.permutation(np.concatenate([hot, rng.choice(num_rows, size=n, replace=False)]))[:n]
            idx = np.unique(idx)[rng.permutation(len(np.unique(idx)))]
        idx = idx.astype(np.int64)
        d_idx = torch.from_numpy(idx).cuda()
        out = torch.full((len(idx), dim), -7.0, dtype=torch.float32, device="cuda")
        cache.read_feature(out.data_ptr(), d_idx.data_ptr(), len(idx))
        got = out.cpu().numpy()
        want = orc.read_feature(idx, oracle.SCHED_HITS_FIRST)
        assert np.array_equal(want, feat[idx])                      # the oracle itself is a pure gather
        assert got.tobytes() == feat[idx].tobytes(), f"row bytes differ at n={n}"
        hit, miss, bad = cache.stats()
        assert (hit, miss, bad) == (orc.hit_cnt, orc.miss_cnt, 0), f"counters differ at n={n}"
        _compare_tables(cache, orc)
        cc = np.zeros(13, dtype=np.int32)
        cache.get_cache_data(cc.ctypes.data, 13)
        assert np.array_equal(cc, orc.color_counters())
    assert orc.hit_cnt > 0 and orc.miss_cnt > 0
    cache.close()
    table.close()


@pytest.mark.parametrize("tag64", [False, True], ids=["tags32", "tags64"])
def test_duplicates_empty_and_set_overflow(hiplib, oracle, torch_cuda, tag64):
    """Duplicate ids inside a batch, an empty batch, and > 32 misses landing in one set within one batch."""
    torch = torch_cuda
    P = hiplib
    dim, num_rows, cache_mb = 1024, 4000, 1  # 8 sets x 32 ways
    feat = oracle.make_features(num_rows, dim, seed=11)
    table = PinnedTable(P, feat)
    cache, _ = _make_cache(P, table, cache_mb, tag64=tag64)
    orc = oracle.OracleCache(cache_mb, dim, feat)
    rng = np.random.default_rng(5)
    batches = [
        np.array([], dtype=np.int64),
        np.array([5, 5, 5, 13, 5, 13, 21], dtype=np.int64),                  # duplicates, all misses first time
        np.array([5, 13, 21, 5], dtype=np.int64),                            # now hits, duplicated
        (np.arange(100, dtype=np.int64) * 8 + 3),                            # 100 ids in ONE set (3): > 32 misses
        (np.arange(100, dtype=np.int64) * 8 + 3)[::-1].copy(),               # same set again, reversed order
        rng.integers(0, num_rows, size=3000).astype(np.int64),               # duplicates at random
    ]
    for idx in batches:
        out = torch.zeros((max(len(idx), 1), dim), dtype=torch.float32, device="cuda")
        d_idx = torch.from_numpy(idx).cuda() if len(idx) else torch.zeros(1, dtype=torch.int64, device="cuda")
        cache.read_feature(out.data_ptr(), d_idx.data_ptr(), len(idx))
        orc.read_feature(idx, oracle.SCHED_HITS_FIRST)
        if len(idx):
            assert out.cpu().numpy()[: len(idx)].tobytes() == feat[idx].tobytes()
        assert cache.stats()[:2] == (orc.hit_cnt, orc.miss_cnt)
        keys, cnt, _ = cache.dump()
        assert np.array_equal(keys, orc.keys()) and np.array_equal(cnt, orc.set_cnt())
    cache.close()
    table.close()


def test_out_of_range_ids_are_rejected(hiplib, oracle, torch_cuda):
    torch = torch_cuda
    P = hiplib
    dim, num_rows = 128, 1000
    feat = oracle.make_features(num_rows, dim, seed=2)
    table = PinnedTable(P, feat)
    cache, _ = _make_cache(P, table, 1)
    idx = np.array([1, 2, num_rows, 3, -1, 2**40, 999], dtype=np.int64)
    out = torch.full((len(idx), dim), 9.0, dtype=torch.float32, device="cuda")
    cache.read_feature(out.data_ptr(), torch.from_numpy(idx).cuda().data_ptr(), len(idx))
    got = out.cpu().numpy()
    ok = np.array([0, 1, 3, 6])
    assert np.array_equal(got[ok], feat[idx[ok]])
    assert np.all(got[[2, 4, 5]] == 0.0)
    hit, miss, bad = cache.stats()
    assert (hit, miss, bad) == (0, 4, 3)
    cache.close()
    table.close()


def test_second_pass_is_all_hits_and_stats_reset(hiplib, oracle, torch_cuda, capsys):
    torch = torch_cuda
    P = hiplib
    dim, num_rows, cache_mb = 1024, 50000, 64  # 512 sets x 32 = 16384 lines
    feat = oracle.make_features(num_rows, dim, seed=9)
    table = PinnedTable(P, feat)
    cache, _ = _make_cache(P, table, cache_mb)
    idx = np.random.default_rng(3).choice(16384, size=8000, replace=False).astype(np.int64)  # <= 32 ids per set
    d_idx = torch.from_numpy(idx).cuda()
    out = torch.empty((len(idx), dim), dtype=torch.float32, device="cuda")
    cache.read_feature(out.data_ptr(), d_idx.data_ptr(), len(idx))
    assert cache.stats() == (0, len(idx), 0)
    out.zero_()
    cache.read_feature(out.data_ptr(), d_idx.data_ptr(), len(idx))
    assert cache.stats() == (len(idx), len(idx), 0)
    assert out.cpu().numpy().tobytes() == feat[idx].tobytes()
    cache.print_stats()  # isolated_cache.h:132-141: prints and resets
    text = capsys.readouterr().out
    assert f"hit count: {len(idx)} miss count: {len(idx)}" in text and "GPU hit ratio: 0.500000" in text
    assert cache.stats() == (0, 0, 0)
    cache.close()
    table.close()


@pytest.mark.parametrize("n,parts", [(0, 4), (1, 1), (5, 2), (255, 3), (256, 8), (257, 8), (10000, 8), (36864, 7), (70001, 16)])
def test_route_and_scatter_match_oracle(hiplib, oracle, torch_cuda, n, parts):
    torch = torch_cuda
    P = hiplib
    dim = 128
    feat = oracle.make_features(64, dim, seed=1)
    table = PinnedTable(P, feat)
    cache, _ = _make_cache(P, table, 1)
    rng = np.random.default_rng(n + parts)
    idx = rng.integers(0, 10**9, size=n).astype(np.int64)
    d_idx = torch.from_numpy(idx).cuda() if n else torch.zeros(1, dtype=torch.int64, device="cuda")
    max_sample = max(n, 1)
    # reference layout [G][max_sample]
    node = torch.full((parts * max_sample,), -1, dtype=torch.int64, device="cuda")
    mp = torch.full((parts * max_sample,), -1, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(parts, dtype=torch.int64, device="cuda")
    cache.split_node_list(d_idx.data_ptr(), n, node.data_ptr(), mp.data_ptr(), cnt.data_ptr(), parts, max_sample)
    o_node, o_map, o_cnt = oracle.split_node_list(idx, parts, max_sample)
    assert np.array_equal(cnt.cpu().numpy(), o_cnt)
    g_node, g_map = node.cpu().numpy(), mp.cpu().numpy()
    for g in range(parts):
        sl = slice(g * max_sample, g * max_sample + int(o_cnt[g]))
        assert np.array_equal(g_node[sl], o_node[sl]) and np.array_equal(g_map[sl], o_map[sl])
    # packed layout + offsets
    node2 = torch.full((max_sample,), -1, dtype=torch.int64, device="cuda")
    map2 = torch.full((max_sample,), -1, dtype=torch.int64, device="cuda")
    cnt2 = torch.zeros(parts, dtype=torch.int64, device="cuda")
    offs = torch.zeros(parts + 1, dtype=torch.int64, device="cuda")
    cache.route(d_idx.data_ptr(), n, parts, node2.data_ptr(), map2.data_ptr(), cnt2.data_ptr(), offs.data_ptr(), 0)
    want_offs = np.concatenate([[0], np.cumsum(o_cnt)])
    assert np.array_equal(offs.cpu().numpy(), want_offs)
    want_node = np.concatenate([o_node[g * max_sample: g * max_sample + int(o_cnt[g])] for g in range(parts)]) if n else np.zeros(0, np.int64)
    want_map = np.concatenate([o_map[g * max_sample: g * max_sample + int(o_cnt[g])] for g in range(parts)]) if n else np.zeros(0, np.int64)
    assert np.array_equal(node2.cpu().numpy()[:n], want_node) and np.array_equal(map2.cpu().numpy()[:n], want_map)
    # scatter (map_feat_data): out[map[r]] = src[r]
    if n:
        src = rng.random((n, dim), dtype=np.float32)
        d_src = torch.from_numpy(src).cuda()
        out = torch.zeros((n, dim), dtype=torch.float32, device="cuda")
        cache.scatter(out.data_ptr(), d_src.data_ptr(), map2.data_ptr(), n)
        want = np.zeros((n, dim), dtype=np.float32)
        oracle.map_feat_data(want, src, want_map)
        assert out.cpu().numpy().tobytes() == want.tobytes()
    cache.close()
    table.close()


def test_full_size_config2_properties(hiplib, torch_cuda):
    """BASELINE config 2 geometry (4 GiB cache, dim 1024, N = 36,864 rows per minibatch) checked through
    size-independent properties: rows equal the procedural table bit for bit, hits + misses == N, a second pass over the
    same ids is all hits, a third pass over fresh ids leaves earlier lines intact (no set holds more than 32 of them)."""
    torch = torch_cuda
    P = hiplib
    from COALA_GNN.synthetic import feature_rows_torch, alloc_pinned_table
    dim, num_rows, cache_mb, n = 1024, 1 << 20, 4096, 36864
    table = alloc_pinned_table(num_rows, dim, seed=5, device=0)
    ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
    cache = P.Isolated_Cache(ctrl, None, 0, 1, cache_mb, table.device_ptr, num_rows=num_rows)
    g = cache.geometry()
    assert (g.num_sets, g.cache_dim, g.line_bytes) == (32768, 1024, 4096)
    gen = torch.Generator(device="cpu").manual_seed(0)
    perm = torch.randperm(num_rows, generator=gen)
    a, b = perm[:n].cuda(), perm[n:2 * n].cuda()
    out = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    for ids, want_stats in ((a, (0, n)), (a, (n, n)), (b, (n, 2 * n)), (a, (2 * n, 2 * n))):
        out.fill_(-1.0)
        cache.read_feature(out.data_ptr(), ids.data_ptr(), n)
        assert torch.equal(out, feature_rows_torch(ids, dim, 5))
        hit, miss, bad = cache.stats()
        assert (hit, miss, bad) == (*want_stats, 0)
    cache.close()
    table.close()


def test_full_size_config2_counters_match_tag_only_oracle(hiplib, oracle, torch_cuda):
    """BASELINE config 2 at full per-minibatch size (4 GiB cache = 32,768 sets, N = 36,864, dim 1024): hit / miss counters
    and the whole tag table equal the oracle's (tag-only mode: no payload on the CPU side) over a sequence of minibatches
    with a hot working set; rows are checked against the procedural table on the GPU."""
    torch = torch_cuda
    P = hiplib
    from COALA_GNN.synthetic import alloc_pinned_table, feature_rows_torch
    dim, num_rows, cache_mb, n = 1024, 3_000_000, 4096, 36864
    table = alloc_pinned_table(num_rows, dim, seed=9, device=0)
    ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
    cache = P.Isolated_Cache(ctrl, None, 0, 1, cache_mb, table.device_ptr, num_rows=num_rows)
    orc = oracle.OracleCache(cache_mb, dim, np.zeros((1, dim), dtype=np.float32), tag_only=True)
    rng = np.random.default_rng(11)
    hot = rng.choice(num_rows, size=60000, replace=False)
    out = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    for step in range(8):
        idx = np.unique(np.concatenate([rng.choice(hot, size=n // 2, replace=False), rng.choice(num_rows, size=n, replace=False)]))
        idx = idx[rng.permutation(len(idx))][:n].astype(np.int64)
        d_idx = torch.from_numpy(idx).cuda()
        cache.read_feature(out.data_ptr(), d_idx.data_ptr(), len(idx))
        orc.read_feature(idx, oracle.SCHED_HITS_FIRST, want_rows=False)
        assert cache.stats()[:2] == (orc.hit_cnt, orc.miss_cnt), f"step {step}"
        assert torch.equal(out[: len(idx)], feature_rows_torch(d_idx, dim, 9))
    keys, cnt, _ = cache.dump()
    assert np.array_equal(keys, orc.keys()) and np.array_equal(cnt, orc.set_cnt())
    assert orc.hit_cnt > 50000
    cache.close()
    table.close()


def test_full_size_config1_products_shape(hiplib, oracle, torch_cuda):
    """BASELINE configs[0] at its STATED size: ogbn-products shape, 2,449,029 x 100 fp32 (0.98 GB cold table; dim 100 < cache_dim 128: the
    stride fix of SURVEY appendix A.2), fan-out 5,5 at bs 1024 through the native sampler on a products-shaped graph (average in-degree 25)
    and COALA_GNN_Manager.fetch_feature (isolated, 128 MiB cache).  The reference runs this configuration on the CPU
    (examples/ssd_gnn_dataloader.py:687-854, DGL sampler + feat[input_nodes]); the CPU side here is the tag-only oracle.  Properties: rows
    bit-equal to the procedural table for every minibatch, hit / miss counters and the final tag table equal the oracle's, input nodes
    unique with the seeds first, and a second pass over one minibatch delivers the same bytes with the oracle's hit count (nearly all hits)."""
    torch = torch_cuda
    from COALA_GNN import MPI_Comm_Manager
    from COALA_GNN.COALA_GNN_Manager import COALA_GNN_Manager
    from COALA_GNN.sampler import NeighborSampler
    from COALA_GNN.synthetic import alloc_pinned_table, feature_rows_torch, powerlaw_csc
    rows, dim, cache_mb, batch, fan = 2_449_029, 100, 128, 1024, [5, 5]
    table = alloc_pinned_table(rows, dim, seed=7, device=0)
    indptr, indices = powerlaw_csc(rows, 25.0, seed=7, device="cuda")
    sampler = NeighborSampler(fan, seed=7)
    g = sampler.make_graph(indptr, indices)
    comm = MPI_Comm_Manager(0)
    comm.initialize_nested_process_group("isolated")
    mgr = COALA_GNN_Manager(None, 1, dim * 4, 1024, 0, cache_mb, batch, fan, dim, comm, "cuda:0", cache_backend="isolated", sim_buf=table,
                            num_rows=rows)
    assert mgr.max_sample_size == 36864
    geo = mgr.COALA_GNN_Cache.geometry()
    assert (geo.cache_dim, geo.line_bytes, geo.num_sets) == (128, 512, oracle.num_sets(cache_mb, 128))
    orc = oracle.OracleCache(cache_mb, dim, np.zeros((1, dim), dtype=np.float32), tag_only=True)
    train = torch.randperm(int(0.6 * rows), generator=torch.Generator().manual_seed(7))
    last = None
    for step in range(40):
        seeds = train[step * batch: (step + 1) * batch].cuda()
        b = sampler.sample(g, seeds)
        ids = b[0]
        assert torch.equal(ids[:batch], seeds) and ids.unique().numel() == ids.numel() and ids.numel() <= mgr.max_sample_size
        got = mgr.fetch_feature(b)[-1]
        assert got.shape == (ids.numel(), dim) and torch.equal(got, feature_rows_torch(ids, dim, 7)), f"step {step}"
        orc.read_feature(ids.cpu().numpy(), oracle.SCHED_HITS_FIRST, want_rows=False)
        assert mgr.COALA_GNN_Cache.stats()[:2] == (orc.hit_cnt, orc.miss_cnt), f"step {step}"
        last = b
    assert orc.hit_cnt > 0 and orc.miss_cnt > 0
    keys, cnt, _ = mgr.COALA_GNN_Cache.dump()
    assert np.array_equal(keys, orc.keys()) and np.array_equal(cnt, orc.set_cnt())
    hit0 = mgr.COALA_GNN_Cache.stats()[0]
    again = mgr.fetch_feature(last)[-1]
    assert torch.equal(again, feature_rows_torch(last[0], dim, 7))
    orc.read_feature(last[0].cpu().numpy(), oracle.SCHED_HITS_FIRST, want_rows=False)
    assert mgr.COALA_GNN_Cache.stats()[:2] == (orc.hit_cnt, orc.miss_cnt)
    # almost all hits -- not all: round-robin eviction does not spare a line that was hit earlier in the same batch (isolated_cache.h:197-210)
    assert mgr.COALA_GNN_Cache.stats()[0] - hit0 >= int(0.9 * last[0].numel())
    del mgr
    table.close()


@pytest.mark.parametrize("name,dim,cache_mb,n,num_rows", [
    ("papers100M 15,10,5 (config 4)", 128, 16384, 1024 * 16 * 11 * 6, 24_000_000),    # 1,081,344 rows per minibatch
    ("IGB-large 10,10,10 (config 5)", 1024, 16384, 1024 * 11 * 11 * 11, 4_000_000),     # 1,362,944 rows = 5.58 GB out
])
def test_full_size_big_configs_properties(hiplib, oracle, torch_cuda, name, dim, cache_mb, n, num_rows):
    """Largest per-minibatch shapes of BASELINE.json (16 GiB caches) through size-independent properties: rows bit-equal
    to the procedural table (checked in slices on the GPU), hits + misses == N, second pass all hits, counters equal the
    tag-only oracle.  Exercises the > 4 GiB output offsets and the 2^20+ row batches."""
    torch = torch_cuda
    P = hiplib
    from COALA_GNN.synthetic import alloc_pinned_table, feature_rows_torch
    table = alloc_pinned_table(num_rows, dim, seed=4, device=0)
    ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
    cache = P.Isolated_Cache(ctrl, None, 0, 1, cache_mb, table.device_ptr, num_rows=num_rows)
    g = cache.geometry()
    assert g.num_sets == oracle.num_sets(cache_mb, oracle.cache_dim(dim))
    orc = oracle.OracleCache(cache_mb, dim, np.zeros((1, dim), dtype=np.float32), tag_only=True)
    ids = torch.randperm(num_rows, generator=torch.Generator().manual_seed(3))[:n]
    d_idx = ids.cuda()
    out = torch.empty((n, dim), dtype=torch.float32, device="cuda")

    def check_rows():
        for lo in range(0, n, 1 << 18):
            hi = min(n, lo + (1 << 18))
            assert torch.equal(out[lo:hi], feature_rows_torch(d_idx[lo:hi], dim, 4))

    for pass_no in range(2):
        out.fill_(-1.0)
        cache.read_feature(out.data_ptr(), d_idx.data_ptr(), n)
        orc.read_feature(ids.numpy(), oracle.SCHED_HITS_FIRST, want_rows=False)
        check_rows()
        hit, miss, bad = cache.stats()
        assert (hit, miss, bad) == (orc.hit_cnt, orc.miss_cnt, 0)
        assert hit + miss == (pass_no + 1) * n
    assert orc.hit_cnt > 0.99 * n  # second pass: (almost) everything was kept (a set overflows only past 32 ids)
    cache.close()
    table.close()


def test_calls_that_change_streams_keep_their_order(hiplib, torch_cuda):
    """A cache handle's tables and scratch are ordered by the stream of its calls; a caller that moves to another (non-blocking)
    stream without synchronising still gets them in program order: the second batch -- the same ids on another stream, right behind
    a 36,864-row cold fill -- must see every line of the first (all hits), the stats read on a third stream both batches."""
    torch = torch_cuda
    P = hiplib
    from COALA_GNN.synthetic import alloc_pinned_table, feature_rows_torch
    dim, num_rows, n = 1024, 1 << 20, 36864
    table = alloc_pinned_table(num_rows, dim, seed=5, device=0)
    ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
    cache = P.Isolated_Cache(ctrl, None, 0, 1, 4096, table.device_ptr, num_rows=num_rows, sync=False)
    ids = torch.randperm(num_rows, generator=torch.Generator().manual_seed(2))[:n].cuda()
    outs = [torch.empty((n, dim), dtype=torch.float32, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for rounds in range(3):
        cache.stats(reset=True)
        fresh = (ids + 1 + rounds) % num_rows if rounds else ids
        with torch.cuda.stream(streams[0]):
            cache.read_feature(outs[0].data_ptr(), fresh.data_ptr(), n)      # ~1.4 ms of PCIe fill (first round: all misses)
        with torch.cuda.stream(streams[1]):
            cache.read_feature(outs[1].data_ptr(), fresh.data_ptr(), n)      # enqueued at once on another stream
        with torch.cuda.stream(streams[2]):
            hit, miss, bad = cache.stats()
        torch.cuda.synchronize()
        assert hit >= n and hit + miss == 2 * n and bad == 0, (rounds, hit, miss)
        want = feature_rows_torch(fresh, dim, 5)
        assert torch.equal(outs[0], want) and torch.equal(outs[1], want)
    cache.close()
    table.close()


def test_fetch_events_ride_on_the_dispatches(hiplib, oracle, torch_cuda):
    """coala_cache_fetch_events: the begin / end events of a read are attached to its first / last kernel launch (no packets of their
    own).  A consumer on ANOTHER stream that waits for the end event (coala_stream_wait_event) must see every row of a cold read --
    20,000 x 4 KiB over PCIe, ~1.5 ms after the call returned -- and begin -> end must be the duration of that read.  n = 0 and a
    profiling handle hand out no events (the caller then records its own)."""
    torch = torch_cuda
    P = hiplib
    dim, num_rows, n = 1024, 60000, 20000
    feat = oracle.make_features(num_rows, dim, seed=21)
    table = PinnedTable(P, feat)
    cache, _ = _make_cache(P, table, 256, sync=False)
    cache.fetch_events(True)
    rng = np.random.default_rng(4)
    producer, consumer = torch.cuda.Stream(), torch.cuda.Stream()
    for step in range(3):           # step 0: all misses; later steps: hits and misses
        ids = rng.choice(num_rows, size=n, replace=False).astype(np.int64)
        idx = torch.from_numpy(ids).cuda()
        out = torch.full((n, dim), -1.0, dtype=torch.float32, device="cuda")
        copy = torch.full((n, dim), -2.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(producer):
            bracket = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            bracket[0].record()
            cache.read_feature(out.data_ptr(), idx.data_ptr(), n)
            bracket[1].record()
        begin, end = cache.last_fetch_events()
        assert begin and end
        with torch.cuda.stream(consumer):
            P.stream_wait_event(end)            # the consumer's (current) stream waits; the host does not
            copy.copy_(out)
        consumer.synchronize()
        assert copy.cpu().numpy().tobytes() == feat[ids].tobytes(), f"step {step}: the consumer ran ahead of the read"
        ms = P.event_elapsed_ms(begin, end, wait=True)
        producer.synchronize()
        outer = bracket[0].elapsed_time(bracket[1])
        assert 0.0 < ms <= outer + 0.05 and ms > 0.5 * outer, (ms, outer)
    cache.read_feature(0, 0, 0)
    assert cache.last_fetch_events() == (None, None)
    cache.fetch_events(False)
    idx = torch.arange(100, device="cuda")
    out = torch.empty((100, dim), dtype=torch.float32, device="cuda")
    cache.read_feature(out.data_ptr(), idx.data_ptr(), 100)
    assert cache.last_fetch_events() == (None, None)
    torch.cuda.synchronize()
    cache.close()
    prof, _ = _make_cache(P, table, 16, profile=True, sync=False)
    prof.fetch_events(True)
    prof.read_feature(out.data_ptr(), idx.data_ptr(), 100)
    assert prof.last_fetch_events() == (None, None)      # the profiling handle uses the dispatches' event slots itself
    torch.cuda.synchronize()
    assert out.cpu().numpy().tobytes() == feat[:100].tobytes()
    prof.close()
    table.close()
