"""Body of the randomized cache sweep (tests/test_fuzz_gpu.py).  Importable (the product library, in-process) and runnable as a script
(`python tests/_fuzz_body.py <seed> <tmpdir>`: the test launches it with COALA_HIP_LIB pointing at the development library and the
fill kernel's launch-shape knobs set, which only that build reads)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "coala-gnn_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(P, oracle, tmp_path, seed, cases=12):
    import torch
    from _util import ColorFiles, PinnedTable
    rng = np.random.default_rng(1000 + seed)
    for case in range(cases):
        dim = int(rng.choice([1, 3, 4, 17, 64, 100, 128, 129, 200, 256, 300, 511, 512, 640, 1000, 1024]))
        cache_mb = int(rng.choice([1, 2, 3, 5]))
        n_gpus = int(rng.choice([1, 1, 2, 3, 8]))
        distributed = n_gpus > 1
        num_rows = int(rng.integers(500, 9000))
        with_color = bool(rng.integers(0, 2))
        num_colors = int(rng.integers(1, 30))
        feat = oracle.make_features(num_rows, dim, seed=case + 10 * seed)
        table = PinnedTable(P, feat)
        color = rng.integers(0, num_colors + 1, size=num_rows).astype(np.int64) if with_color else None
        nd = None
        if with_color:
            d = os.path.join(str(tmp_path), f"s{seed}c{case}")
            os.makedirs(d, exist_ok=True)
            files = ColorFiles(d, color, np.zeros((num_colors, 2), np.int64), np.zeros((num_colors, 2)))
            items = np.zeros(2, dtype=np.int64)
            nd = P.Node_distributor_pybind(items.ctypes.data, 0, 1, 1, 1, files.color_file, files.topk_file, files.score_file)
        ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
        cls = P.SSD_GNN_NVSHMEM_Cache if distributed else P.Isolated_Cache
        tag64 = bool(rng.integers(0, 2))   # the reference's 64-bit tags, or one 128-B line of 32-bit tags per set (the default)
        cache = cls(ctrl, nd, 0, n_gpus, cache_mb, table.device_ptr, num_rows=num_rows, rank=0, tag64=tag64)
        orc = oracle.OracleCache(cache_mb, dim, feat, node_color=color, num_colors=num_colors if with_color else 0, n_gpus=n_gpus,
                                 distributed=distributed)
        bad_total = 0
        for b in range(5):
            kind = rng.integers(0, 5)
            n = int(rng.integers(0, 3000)) if kind else 0
            if kind == 1:
                idx = rng.integers(0, num_rows, size=n)                                   # duplicates
            elif kind == 2:
                idx = (rng.integers(0, max(num_rows // max(orc.num_sets, 1), 1), size=n) * orc.num_sets * (n_gpus if distributed else 1)) % num_rows  # one set
            else:
                idx = rng.choice(num_rows, size=min(n, num_rows), replace=False)
            idx = idx.astype(np.int64)
            bad = np.zeros(len(idx), dtype=bool)
            if kind == 4 and len(idx):
                bad = rng.random(len(idx)) < 0.05
                idx[bad] = rng.choice([-1, num_rows, num_rows + 7, 2**40], size=int(bad.sum()))
            good = ~bad
            n = len(idx)
            d_idx = torch.from_numpy(idx).cuda() if n else torch.zeros(1, dtype=torch.int64, device="cuda")
            out = torch.full((max(n, 1), dim), -2.0, dtype=torch.float32, device="cuda")
            split = distributed and n > 0 and rng.random() < 0.6
            if not split:
                (cache.serve if distributed else cache.read_feature)(out.data_ptr(), d_idx.data_ptr(), n)
                got = out.cpu().numpy()[:n]
            else:
                # the split serve of the distributed fetch: a random slice of the batch is redirected through a random row map into
                # another tensor, the fills arrive as random range sets in random order (one batch for the table all the same)
                lo, hi = sorted(int(x) for x in rng.integers(0, n + 1, size=2))
                perm = rng.permutation(hi - lo).astype(np.int64)
                use_map = rng.random() < 0.7
                d_map = torch.from_numpy(perm).cuda() if (use_map and hi > lo) else None
                other = torch.full((max(hi - lo, 1), dim), -4.0, dtype=torch.float32, device="cuda")
                cache.serve_probe_redirect(out.data_ptr(), d_idx.data_ptr(), n, lo, hi, other.data_ptr(), d_map.data_ptr() if d_map is not None else 0)
                cuts = sorted({0, n, *(int(c) for c in rng.integers(0, n + 1, size=int(rng.integers(0, 9))))})
                ranges = list(zip(cuts[:-1], cuts[1:]))
                rng.shuffle(ranges)
                k = int(rng.integers(1, 4))
                for part in range(k):
                    if ranges[part::k]:      # (a fill after the batch has been covered is refused: nothing left to match it)
                        cache.serve_fill_ranges(out.data_ptr(), d_idx.data_ptr(), n, ranges[part::k])
                got = out.cpu().numpy()[:n].copy()
                oth = other.cpu().numpy()
                inside = np.zeros(n, dtype=bool)
                inside[lo:hi] = True
                assert np.all(got[inside] == -2.0), "a redirected row was also written to the batch's own output"
                rows_of = perm if d_map is not None else np.arange(hi - lo)
                got[lo:hi] = oth[rows_of]
            orc.read_feature(idx[good], oracle.SCHED_HITS_FIRST, want_rows=False)      # the oracle never sees rejected ids
            assert got[good].tobytes() == feat[idx[good]].tobytes(), f"seed {seed} case {case} batch {b} (split={split})"
            assert np.all(got[bad] == 0.0)
            bad_total += int(bad.sum())
            assert cache.stats() == (orc.hit_cnt, orc.miss_cnt, bad_total), f"seed {seed} case {case} batch {b} (dim {dim}, G {n_gpus}, split={split})"
            keys, cnt, meta = cache.dump()
            assert np.array_equal(keys, orc.keys()) and np.array_equal(cnt, orc.set_cnt())
            if with_color:
                cc = np.zeros(num_colors + 1, dtype=np.int32)
                cache.get_cache_data(cc.ctypes.data, num_colors + 1)
                assert np.array_equal(cc, orc.color_counters()) and np.array_equal(meta.astype(np.uint64), orc.color_meta())
        cache.close()
        table.close()


if __name__ == "__main__":
    import COALA_GNN_Pybind as P
    from oracle import oracle as O
    O.build()
    run(P, O, sys.argv[2], int(sys.argv[1]))
    print(f"fuzz seed {sys.argv[1]} ok ({os.environ.get('COALA_HIP_LIB', 'product library')})")
