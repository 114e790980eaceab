#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the CPU oracle (oracle/coala_oracle.c).

The reference ships no fixtures and cannot be built or imported here, so these vectors pin the
build against ITSELF: (a) the oracle against regressions, (b) the HIP path against the oracle at fixed seeds.  The
independent truth in every cache fixture is rows == feat[idx] (recomputable from the procedural feature formula).
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def cache_case(name, dim, cache_mb, num_rows, n_gpus, distributed, num_colors, seed, batch_sizes):
    rng = np.random.default_rng(seed)
    feat = O.make_features(num_rows, dim, seed=seed)
    color = rng.integers(0, num_colors + 1, size=num_rows).astype(np.int64)
    orc = O.OracleCache(cache_mb, dim, feat, node_color=color, num_colors=num_colors, n_gpus=n_gpus, distributed=distributed)
    hot = rng.choice(num_rows, size=min(num_rows, 600), replace=False)
    batches, steps = [], []
    for n in batch_sizes:
        cold = rng.choice(num_rows, size=n, replace=False)
        idx = np.unique(np.concatenate([hot[: n // 2], cold]))[:n]
        idx = idx[rng.permutation(len(idx))].astype(np.int64)
        rows = orc.read_feature(idx, O.SCHED_HITS_FIRST)
        assert np.array_equal(rows, feat[idx])
        batches.append(idx)
        steps.append({"n": int(len(idx)), "hit": orc.hit_cnt, "miss": orc.miss_cnt, "rows_sha256": sha(rows),
                      "keys_sha256": sha(orc.keys()), "set_cnt_sha256": sha(orc.set_cnt()),
                      "color_counters": orc.color_counters().tolist()})
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), color=color, **{f"idx{i}": b for i, b in enumerate(batches)})
    return {"name": name, "dim": dim, "cache_mb": cache_mb, "num_rows": num_rows, "n_gpus": n_gpus,
            "distributed": distributed, "num_colors": num_colors, "feat_seed": seed, "num_sets": orc.num_sets,
            "cache_dim": orc.cache_dim, "steps": steps}


def distributor_case(name, num_nodes, batch, local_size, num_ids, num_colors, topk, seed):
    rng = np.random.default_rng(seed)
    color = rng.integers(0, num_colors + 1, size=num_ids).astype(np.int64)
    tk = rng.integers(0, num_colors + 1, size=(num_colors, topk)).astype(np.int64)
    sc = rng.random((num_colors, topk))
    items = rng.permutation(num_ids).astype(np.int64)
    meta = [rng.integers(0, 50, size=num_colors + 1).astype(np.int32) for _ in range(num_nodes)]
    for m in meta:
        m[rng.random(num_colors + 1) < 0.3] = 0
    outs = []
    glob = batch * local_size * num_nodes
    for off in (0, glob):
        outs.append([O.distribute_node_with_affinity(items, off, batch, local_size, j, num_nodes, color, tk, sc, meta).tolist()
                     for j in range(num_nodes)])
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), color=color, topk=tk, score=sc, items=items,
                        **{f"meta{j}": m for j, m in enumerate(meta)})
    return {"name": name, "num_nodes": num_nodes, "batch": batch, "local_size": local_size, "offsets": [0, glob], "out": outs}


def sampler_case(name, n_nodes, avg_deg, fanouts, n_seeds, seed):
    rng = np.random.default_rng(seed)
    deg = np.minimum(rng.geometric(1.0 / avg_deg, size=n_nodes), 200)
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    indices = rng.integers(0, n_nodes, size=int(indptr[-1])).astype(np.int64)
    seeds = rng.choice(n_nodes, size=n_seeds, replace=False).astype(np.int64)
    layers = O.sample_blocks(indptr, indices, seeds, list(reversed(fanouts)), 99, 3)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), indptr=indptr, indices=indices, seeds=seeds)
    return {"name": name, "fanouts": fanouts, "rng_seed": 99, "step": 3,
            "layers": [{"n_src": int(len(s)), "src_sha256": sha(s), "local_sha256": sha(l)} for s, l, _ in layers]}


def main():
    gold = {"cache": [
        cache_case("cache_igb_iso", 1024, 1, 6000, 1, False, 12, 3, [1, 64, 257, 1000, 3000, 999]),
        cache_case("cache_papers_iso", 128, 1, 20000, 1, False, 7, 5, [3, 500, 4097, 2500]),
        cache_case("cache_products_iso", 100, 1, 20000, 1, False, 7, 6, [65, 2000, 4000]),
        cache_case("cache_igb_dist4", 1024, 1, 8000, 4, True, 12, 7, [1000, 2500, 2500]),
    ], "distributor": [
        distributor_case("dist_1dom", 1, 16, 2, 4000, 20, 10, 1),
        distributor_case("dist_2dom", 2, 16, 2, 4000, 20, 10, 2),
        distributor_case("dist_4dom", 4, 8, 4, 4000, 20, 10, 3),
    ], "sampler": [
        sampler_case("sampler_55", 5000, 8.0, [5, 5], 256, 4),
        sampler_case("sampler_1055", 5000, 12.0, [10, 5, 5], 64, 5),
        # the fan-outs of BASELINE.json configs[2], [4] and [3]
        sampler_case("sampler_1010", 6000, 12.0, [10, 10], 128, 6),
        sampler_case("sampler_101010", 6000, 12.0, [10, 10, 10], 48, 7),
        sampler_case("sampler_15105", 6000, 14.5, [15, 10, 5], 64, 8),
    ], "feat_values": {f"{r},{c},{s}": float(O.feat_value(r, c, s)) for r, c, s in
                       [(0, 0, 0), (1, 0, 0), (0, 1, 0), (123456, 1023, 7), (99999999, 127, 1), (2**31, 5, 9)]}}
    with open(os.path.join(OUT, "golden.json"), "w") as f:
        json.dump(gold, f, indent=1)
    print("wrote", os.path.join(OUT, "golden.json"))


if __name__ == "__main__":
    main()
