"""Worker of the world_size>1 CPU tests (gloo).  Launched by tests/test_dist_gloo_cpu.py with RANK/WORLD_SIZE/MASTER_* set.

The device primitives (route / serve / scatter) are provided by a TEST DOUBLE built on the oracle, so that the host logic
of the N>1 path -- topology, count exchange, all-to-all-v splits, un-permute bookkeeping, seed scheduling -- runs
without a GPU.  The product never selects this double: COALA_GNN_Manager always binds the HIP cache objects."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "coala-gnn_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import oracle as O  # noqa: E402


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(int(ptr))
    return np.frombuffer(buf, dtype=dtype)


class OracleOps:
    """route / serve / scatter with the signatures of the COALA_GNN_Pybind cache objects, on host memory."""

    def __init__(self, cache, dim):
        self.cache, self.dim = cache, dim
        self._open = None

    def route(self, idx_ptr, n, n_parts, node_ptr, map_ptr, counts_ptr, offsets_ptr=0, bucket_stride=0):
        idx = _arr(idx_ptr, n, np.int64)
        node, mp, cnt = O.split_node_list(idx, n_parts, max(n, 1))
        offs = np.concatenate([[0], np.cumsum(cnt)])
        on, om = _arr(node_ptr, max(n, 1), np.int64), _arr(map_ptr, max(n, 1), np.int64)
        for g in range(n_parts):
            on[offs[g]: offs[g + 1]] = node[g * max(n, 1): g * max(n, 1) + cnt[g]]
            om[offs[g]: offs[g + 1]] = mp[g * max(n, 1): g * max(n, 1) + cnt[g]]
        _arr(counts_ptr, n_parts, np.int64)[:] = cnt
        if offsets_ptr:
            _arr(offsets_ptr, n_parts + 1, np.int64)[:] = offs

    def serve(self, out_ptr, ids_ptr, n):
        if n:
            _arr(out_ptr, n * self.dim, np.float32).reshape(n, self.dim)[:] = self.cache.read_feature(_arr(ids_ptr, n, np.int64), O.SCHED_HITS_FIRST)

    # split-phase serve: the double computes the whole batch at the probe (one oracle batch, like the kernels' K1 over the whole
    # batch) but DELIVERS a row only when its position is filled -- an exchange that ships a slice before its fill fails here
    def serve_probe_redirect(self, out_ptr, ids_ptr, n, begin, end, redirect_out_ptr, row_map_ptr=0):
        assert getattr(self, "_open", None) is None, "probe while a batch is open"
        rows = self.cache.read_feature(_arr(ids_ptr, n, np.int64), O.SCHED_HITS_FIRST)
        self._open = dict(rows=rows, n=n, out=out_ptr, begin=begin, end=end, rout=redirect_out_ptr, rmap=row_map_ptr, filled=np.zeros(n, bool))

    def serve_fill_ranges(self, out_ptr, ids_ptr, n, ranges):
        o = self._open
        assert o is not None and o["n"] == n and o["out"] == out_ptr
        out = _arr(out_ptr, n * self.dim, np.float32).reshape(n, self.dim)
        for b, e in ranges:
            assert 0 <= b <= e <= n and not o["filled"][b:e].any(), "fill ranges overlap"
            o["filled"][b:e] = True
            for pos in range(b, e):
                if o["begin"] <= pos < o["end"]:
                    k = pos - o["begin"]
                    row = int(_arr(o["rmap"] + 8 * k, 1, np.int64)[0]) if o["rmap"] else k
                    _arr(o["rout"] + row * self.dim * 4, self.dim, np.float32)[:] = o["rows"][pos]
                else:
                    out[pos] = o["rows"][pos]
        if o["filled"].all():
            self._open = None

    def scatter_ranges(self, out_ptr, src_ptr, map_ptr, ranges):
        for b, e in ranges:
            for r in range(b, e):
                dst = int(_arr(map_ptr + 8 * r, 1, np.int64)[0])
                _arr(out_ptr + dst * self.dim * 4, self.dim, np.float32)[:] = _arr(src_ptr + r * self.dim * 4, self.dim, np.float32)

    def scatter(self, out_ptr, src_ptr, map_ptr, n):
        if n:
            out = _arr(out_ptr, n * self.dim, np.float32).reshape(n, self.dim)
            O.map_feat_data(out, _arr(src_ptr, n * self.dim, np.float32).reshape(n, self.dim), _arr(map_ptr, n, np.int64))


def mode_exchange():
    from COALA_GNN import MPI_Comm_Manager
    from COALA_GNN.COALA_GNN_Manager import AllToAllExchange
    comm = MPI_Comm_Manager(0)
    comm.initialize_nested_process_group("nccl")
    G, r = comm.local_size, comm.local_rank
    assert (G, comm.global_size, comm.is_master) == (int(os.environ["WORLD_SIZE"]), G, r == 0)
    assert comm.nccl_cache_gather is not None and comm.master_process_list == [0]
    dim, rows = 24, 6000
    feat = O.make_features(rows, dim, seed=5)
    mine = O.OracleCache(1, dim, feat, n_gpus=G, distributed=True)
    ref = [O.OracleCache(1, dim, feat, n_gpus=G, distributed=True) for _ in range(G)]  # single-process reference of all owners
    ex = AllToAllExchange(comm.nccl_cache_gather, r, G, dim, "cpu")
    ops = OracleOps(mine, dim)
    for step in range(5):
        rng = np.random.default_rng(100 + step)  # same stream on every rank: each rank knows every rank's batch
        lists = [rng.choice(rows, size=int(rng.integers(0, 1500)) if step != 2 or g else 0, replace=False).astype(np.int64) for g in range(G)]
        idx = torch.from_numpy(lists[r].copy())
        out = torch.full((max(len(idx), 1), dim), -1.0)
        ex.rounds = 1 + step % 3                      # the same on every rank
        ex.fetch(ops, out.data_ptr(), idx.data_ptr(), len(idx))
        assert ops._open is None, "the exchange left a batch open"
        want = O.dist_fetch(ref, lists)
        assert np.array_equal(out.numpy()[: len(idx)], feat[lists[r]]), f"rank {r} step {step}: rows differ"
        assert np.array_equal(want[r], feat[lists[r]])
        assert (mine.hit_cnt, mine.miss_cnt) == (ref[r].hit_cnt, ref[r].miss_cnt), f"rank {r}: owner counters differ from the collective oracle"
        assert np.array_equal(mine.keys(), ref[r].keys())
        assert sum(ex.last_send_counts) == len(idx) and ex.last_send_counts == [int((lists[r] % G == g).sum()) for g in range(G)]
    # the bucketed fast path: ids pre-bucketed by owner (what the sampler delivers), counts on the "device"
    for step in range(5, 8):
        rng = np.random.default_rng(100 + step)
        raw = [rng.choice(rows, size=int(rng.integers(0, 1500)) if step != 6 or g else 0, replace=False).astype(np.int64) for g in range(G)]
        lists = [np.concatenate([x[x % G == o] for o in range(G)]) if len(x) else x for x in raw]   # stable partition
        idx = torch.from_numpy(lists[r].copy())
        cnt = torch.tensor([int((lists[r] % G == o).sum()) for o in range(G)], dtype=torch.int64)
        out = torch.full((max(len(idx), 1), dim), -1.0)
        ex.rounds = 1 + step % 3
        ex.fetch_bucketed(ops, out.data_ptr(), idx.data_ptr(), len(idx), cnt.data_ptr())
        assert ops._open is None
        O.dist_fetch(ref, lists)
        assert np.array_equal(out.numpy()[: len(idx)], feat[lists[r]]), f"rank {r} step {step}: bucketed rows differ"
        assert (mine.hit_cnt, mine.miss_cnt) == (ref[r].hit_cnt, ref[r].miss_cnt) and np.array_equal(mine.keys(), ref[r].keys())
    comm.destroy_process_group()


class _FakeCache:
    def __init__(self, rank, n):
        self.rank, self.n, self.calls = rank, n, 0

    def get_cache_data(self, ptr, n_entries=None):
        self.calls += 1
        a = _arr(ptr, self.n, np.int32)
        a[:] = (np.arange(self.n) * (self.rank + 1) + self.calls) % 7


class _FakeManager:
    def __init__(self, rank, n):
        self.COALA_GNN_Cache = _FakeCache(rank, n)


def mode_scheduler(two_domains, ranks_per_domain=1):
    """Seed distribution pipeline (COALA_GNN_DataLoader.py:27-75, Training_node_distributor.py:40-60).  ranks_per_domain > 1 with two_domains: 2 domains
    x K ranks (world 2K) -- the shape of bench.py's colour-affinity leg at N = 2K: the domain master parses, the domain's ranks share the result."""
    from _util import ColorFiles, synth_colors
    from COALA_GNN import MPI_Comm_Manager, Node_Distributor
    from COALA_GNN.COALA_GNN_DataLoader import COALA_GNN_Node_Distribution_Scheduler
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    K = ranks_per_domain
    comm = MPI_Comm_Manager(rank // K if two_domains else 0)
    comm.initialize_nested_process_group("isolated")
    if two_domains and K > 1:
        assert (comm.local_size, comm.num_master_process, comm.is_master, comm.master_process_index, comm.local_rank) == (K, world // K, rank % K == 0, rank // K, rank % K)
    elif two_domains:
        assert (comm.local_size, comm.num_master_process, comm.is_master, comm.master_process_index) == (1, world, True, rank)
    else:
        assert (comm.local_size, comm.num_master_process, comm.local_rank) == (world, 1, rank)
    tmp = os.environ["COALA_TEST_TMP"]
    n_ids, batch, ncol = 4000, 16, 10
    color, tk, sc = synth_colors(n_ids, ncol, seed=7)
    files = ColorFiles(tmp, color, tk, sc) if rank == 0 else None
    dist.barrier()
    if files is None:
        files = type("F", (), {"color_file": os.path.join(tmp, "color.npy"), "topk_file": os.path.join(tmp, "topk.npy"),
                               "score_file": os.path.join(tmp, "score.npy")})()
    ids = torch.from_numpy(np.random.default_rng(3).permutation(n_ids).astype(np.int64))
    for method in ("baseline", "node_color"):
        nd = Node_Distributor(comm, ids, batch, files.color_file, files.topk_file, files.score_file, parsing_method=method)
        assert nd.num_colors == ncol and nd.global_batch_size == batch * world
        sched = COALA_GNN_Node_Distribution_Scheduler(nd, _FakeManager(rank, ncol + 1), refresh_counter=2)
        seen = []
        for step in range(7):
            seeds = sched.run(step == 6)
            assert seeds.shape == (batch,) and seeds.dtype == torch.int64
            seen.append(seeds.numpy().copy())
        sched.drain()
        got = np.stack(seen)
        allg = [torch.zeros_like(torch.from_numpy(got)) for _ in range(world)]
        dist.all_gather(allg, torch.from_numpy(got))
        for step in range(7):
            union = np.sort(np.concatenate([a.numpy()[step] for a in allg]))
            want = np.sort(ids.numpy()[step * batch * world: (step + 1) * batch * world])
            assert np.array_equal(union, want), f"{method} step {step}: the global batch is not partitioned exactly"
        if method == "baseline" or not two_domains:
            lo = comm.master_process_index * nd.domain_batch_size + comm.local_rank * batch
            for step in range(7):
                assert np.array_equal(got[step], ids.numpy()[step * batch * world + lo: step * batch * world + lo + batch])
        if two_domains and method == "node_color":
            # replay with the oracle: same colour-count history as the scheduler saw (double-buffered, refreshed every 2 steps)
            assert sched.ssd_gnn_manager.COALA_GNN_Cache.calls >= 2
    comm.destroy_process_group()


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "exchange":
        mode_exchange()
    elif mode == "sched1":
        mode_scheduler(False)
    elif mode == "sched2":
        mode_scheduler(True)
    elif mode == "sched2x2":
        mode_scheduler(True, ranks_per_domain=2)
    else:
        raise SystemExit("unknown mode")
    print(f"rank {os.environ['RANK']} ok")
