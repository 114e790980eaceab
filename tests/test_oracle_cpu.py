"""CPU tests of the oracle itself: the C restatement against an independent numpy mirror, the committed golden vectors,
and the mathematical truth rows == feat[idx].  (The reference has no tests or fixtures: parity unpinned, see
oracle/coala_oracle.h.)"""
import hashlib
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLD, "golden.json")) as f:
        return json.load(f)


def test_geometry_rules(oracle):
    # ssd_gnn_cache.cuh:34-47,96-97
    assert [oracle.cache_dim(d) for d in (1, 100, 128, 129, 256, 300, 512, 513, 1024)] == [128, 128, 128, 256, 256, 512, 512, 1024, 1024]
    assert oracle.cache_dim(1025) == -1
    assert oracle.num_sets(4096, 1024) == 32768
    assert oracle.num_sets(16384, 128) == 1048576     # config 4
    assert oracle.num_sets(16384, 1024) == 131072     # config 5


def test_feature_formula(oracle, golden):
    for k, v in golden["feat_values"].items():
        r, c, s = (int(x) for x in k.split(","))
        assert oracle.feat_value(r, c, s) == v
    f = oracle.make_features(50, 100, seed=7, row0=1000)
    for r, c in ((0, 0), (3, 99), (49, 50)):
        assert f[r, c] == oracle.feat_value(1000 + r, c, 7)
    assert f.min() >= 0.0 and f.max() < 1.0


@pytest.mark.parametrize("sched", [0, 1])
@pytest.mark.parametrize("dist,g", [(False, 1), (True, 4)])
def test_c_oracle_matches_numpy_mirror(oracle, sched, dist, g):
    rng = np.random.default_rng(10 * sched + g)
    dim, rows = 64, 3000
    feat = oracle.make_features(rows, dim, seed=1)
    color = rng.integers(0, 9, size=rows)
    c = oracle.OracleCache(1, dim, feat, node_color=color, num_colors=8, n_gpus=g, distributed=dist)
    m = oracle.PyMirrorCache(c.num_sets, dim, feat, node_color=color, num_colors=8, n_gpus=g, distributed=dist)
    for n in (1, 50, 700, 2999, 10):
        idx = rng.integers(0, rows, size=n) if n == 700 else rng.choice(rows, size=n, replace=False)  # one batch with duplicates
        a, b = c.read_feature(idx, sched), m.read_feature(idx, sched)
        assert np.array_equal(a, feat[idx]) and np.array_equal(b, feat[idx])
        assert (c.hit_cnt, c.miss_cnt) == (m.hit, m.miss)
        assert np.array_equal(c.keys(), m.keys) and np.array_equal(c.set_cnt(), m.set_cnt)
        assert np.array_equal(c.color_meta(), m.color_meta) and np.array_equal(c.color_counters(), m.color_counters)


def test_schedules_agree_on_rows_and_differ_only_in_counters(oracle):
    """Both linearisations of the reference's independent warps deliver the same rows; with unique ids per batch (DGL
    input nodes) a later row can never hit a line inserted earlier in the same batch, so the only difference is an
    eviction that lands before (sequential) or after (hits-first) a later row's lookup."""
    rng = np.random.default_rng(0)
    feat = oracle.make_features(4000, 32, seed=2)
    a = oracle.OracleCache(1, 32, feat)
    b = oracle.OracleCache(1, 32, feat)
    for _ in range(6):
        idx = rng.choice(4000, size=1500, replace=False)
        assert np.array_equal(a.read_feature(idx, 0), b.read_feature(idx, 1))
        assert a.hit_cnt + a.miss_cnt == b.hit_cnt + b.miss_cnt
    assert b.hit_cnt >= a.hit_cnt  # hits-first never loses a hit to an in-batch eviction


def test_first_touch_drives_colour_zero_negative(oracle):
    # isolated_cache.h:427-429 with color_meta zero-initialised (:563): SURVEY.md section 3.3
    feat = oracle.make_features(100, 8, seed=0)
    color = np.full(100, 3)
    c = oracle.OracleCache(1, 8, feat, node_color=color, num_colors=5)
    c.read_feature(np.arange(10))
    cc = c.color_counters()
    assert cc[0] == -10 and cc[3] == 10 and cc.sum() == 0


def test_cache_golden_vectors(oracle, golden):
    for case in golden["cache"]:
        data = np.load(os.path.join(GOLD, case["name"] + ".npz"))
        feat = oracle.make_features(case["num_rows"], case["dim"], seed=case["feat_seed"])
        orc = oracle.OracleCache(case["cache_mb"], case["dim"], feat, node_color=data["color"], num_colors=case["num_colors"],
                                 n_gpus=case["n_gpus"], distributed=case["distributed"])
        assert (orc.num_sets, orc.cache_dim) == (case["num_sets"], case["cache_dim"])
        for i, st in enumerate(case["steps"]):
            idx = data[f"idx{i}"]
            rows = orc.read_feature(idx, oracle.SCHED_HITS_FIRST)
            assert sha(rows) == st["rows_sha256"] == sha(feat[idx])
            assert (orc.hit_cnt, orc.miss_cnt) == (st["hit"], st["miss"])
            assert sha(orc.keys()) == st["keys_sha256"] and sha(orc.set_cnt()) == st["set_cnt_sha256"]
            assert orc.color_counters().tolist() == st["color_counters"]


def test_split_and_map_roundtrip(oracle):
    rng = np.random.default_rng(4)
    idx = rng.integers(0, 10**6, size=5000).astype(np.int64)
    for G in (1, 2, 3, 8):
        node, mp, cnt = oracle.split_node_list(idx, G, 5000)
        assert cnt.sum() == 5000
        src = rng.random((5000, 4), dtype=np.float32)
        packed_map = np.concatenate([mp[g * 5000: g * 5000 + cnt[g]] for g in range(G)])
        for g in range(G):
            part = node[g * 5000: g * 5000 + cnt[g]]
            assert np.all(part % G == g)                                 # cache_kernel.cu:86
            assert np.all(np.diff(mp[g * 5000: g * 5000 + cnt[g]]) > 0)  # stable inside a bucket
            assert np.array_equal(idx[mp[g * 5000: g * 5000 + cnt[g]]], part)
        out = np.zeros((5000, 4), dtype=np.float32)
        oracle.map_feat_data(out, src, packed_map)
        assert np.array_equal(out[packed_map], src)                      # cache_kernel.cu:129-137


def test_dist_fetch_is_a_pure_gather_and_owner_partitioned(oracle):
    rng = np.random.default_rng(9)
    G, dim, rows = 4, 16, 5000
    feat = oracle.make_features(rows, dim, seed=3)
    caches = [oracle.OracleCache(1, dim, feat, n_gpus=G, distributed=True) for _ in range(G)]
    for _ in range(4):
        idx = [rng.choice(rows, size=int(rng.integers(0, 900)), replace=False).astype(np.int64) for _ in range(G)]
        outs = oracle.dist_fetch(caches, idx)
        for g in range(G):
            assert np.array_equal(outs[g], feat[idx[g]])
    for g, c in enumerate(caches):
        k = c.keys()
        live = k[k != np.uint64(oracle.EMPTY_KEY)]
        assert len(live) and np.all(live % np.uint64(G) == np.uint64(g))


def test_sampler_golden_and_properties(oracle, golden):
    for case in golden["sampler"]:
        d = np.load(os.path.join(GOLD, case["name"] + ".npz"))
        layers = oracle.sample_blocks(d["indptr"], d["indices"], d["seeds"], list(reversed(case["fanouts"])), case["rng_seed"], case["step"])
        dst = d["seeds"]
        for (src, local, nbr), want, f in zip(layers, case["layers"], reversed(case["fanouts"])):
            assert len(src) == want["n_src"] and sha(src) == want["src_sha256"] and sha(local) == want["local_sha256"]
            assert np.array_equal(src[: len(dst)], dst) and len(np.unique(src)) == len(src)
            deg = d["indptr"][dst + 1] - d["indptr"][dst]
            assert np.array_equal((local >= 0).sum(1), np.minimum(deg, f))
            assert np.array_equal(src[local[local >= 0]], nbr[nbr >= 0])
            dst = src


@pytest.mark.parametrize("threads", [2, 3, 8])
def test_sampler_twin_is_thread_count_independent(oracle, golden, threads):
    """The all-core CPU sampler of bench.py's cpu_baseline (OpenMP draws + CAS/prefix-sum compaction) returns exactly the blocks of
    the one-core twin, golden vectors included, also with duplicated and out-of-range seeds."""
    for case in golden["sampler"]:
        d = np.load(os.path.join(GOLD, case["name"] + ".npz"))
        args = (d["indptr"], d["indices"], d["seeds"], list(reversed(case["fanouts"])), case["rng_seed"], case["step"])
        for a, b, want in zip(oracle.sample_blocks(*args), oracle.sample_blocks(*args, threads=threads), case["layers"]):
            assert all(np.array_equal(x, y) for x, y in zip(a, b)) and sha(b[0]) == want["src_sha256"]
    rng = np.random.default_rng(4)
    n = 5000
    deg = rng.integers(0, 30, size=n)
    ip = np.zeros(n + 1, dtype=np.int64)
    ip[1:] = np.cumsum(deg)
    ix = rng.integers(0, n, size=int(ip[-1])).astype(np.int64)
    seeds = rng.integers(-2, n + 2, size=700).astype(np.int64)       # duplicates, -1/-2 and n, n+1 (rejected: rows of -1)
    for a, b in zip(oracle.sample_blocks(ip, ix, seeds, [7, 3], 9, 2), oracle.sample_blocks(ip, ix, seeds, [7, 3], 9, 2, threads=threads)):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_openmp_row_gather_is_a_gather(oracle, threads):
    """orc_gather_rows_mt (bench.py's cpu_baseline leg only): out[i] = table[idx[i]] for any thread count, duplicates and an empty batch included."""
    rng = np.random.default_rng(3)
    table = rng.random((5000, 100), dtype=np.float32)
    idx = rng.integers(0, 5000, size=3333).astype(np.int64)
    out = np.full((4000, 100), -1.0, dtype=np.float32)
    used = oracle.gather_rows_mt(table, idx, out, threads)
    assert 1 <= used <= threads
    assert out[:3333].tobytes() == table[idx].tobytes() and (out[3333:] == -1.0).all()
    assert oracle.gather_rows_mt(table, idx[:0], out, threads) >= 1 and out[:3333].tobytes() == table[idx].tobytes()
