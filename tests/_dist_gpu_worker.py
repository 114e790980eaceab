"""Multi-process run of the owner-partitioned path on a ONE-GPU box: every rank drives GPU 0 with its own HIP cache handle
and its own shard of the cold tier; ids and rows cross processes through AllToAllExchange over a gloo group with host
staging (RCCL cannot place two ranks on one device).  Checked against the oracle's collective step."""
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


def main():
    backend_name = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    real = os.environ.get("COALA_TEST_REAL_RCCL") == "1"     # one rank per GPU, RCCL cache group, the fused native exchange
    dev = rank if real else 0
    if real:
        dist.init_process_group("cpu:gloo,cuda:nccl", rank=rank, world_size=world)
    else:
        os.environ["COALA_CACHE_GROUP_BACKEND"] = "gloo"
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(dev)
    from COALA_GNN import MPI_Comm_Manager
    from COALA_GNN.COALA_GNN_Manager import COALA_GNN_Manager
    from _util import PinnedTable
    import COALA_GNN_Pybind as P
    comm = MPI_Comm_Manager(0)
    comm.device_index = dev                     # the one-GPU variant: every rank drives GPU 0
    comm.initialize_nested_process_group(backend_name)
    assert (comm.local_size, comm.local_rank) == (world, rank)
    dim, rows, cache_mb = 256, 30000, 2
    feat = O.make_features(rows, dim, seed=13)
    shard = PinnedTable(P, np.ascontiguousarray(feat[rank::world]), device=dev)   # owner-partitioned cold tier
    mgr = COALA_GNN_Manager(None, 1, dim * 4, 1024, 0, cache_mb, 64, [5, 5], dim, comm, f"cuda:{dev}", cache_backend=backend_name,
                            sim_buf=shard, num_rows=rows, cold_partitioned=True)
    if real:
        assert mgr.exchange_kind == "native" and mgr.exchange.rccl_ranks == world
    else:
        assert mgr.exchange.stage_through_host
    ref = [O.OracleCache(cache_mb, dim, feat, n_gpus=world, distributed=True) for _ in range(world)]
    for step in range(6):
        rng = np.random.default_rng(500 + step)
        lists = [rng.choice(rows // 2, size=int(rng.integers(1, 2304)) if not (step == 3 and g == 1) else 0, replace=False).astype(np.int64)
                 for g in range(world)]
        idx = torch.from_numpy(lists[rank].copy()).to(f"cuda:{dev}")
        out = mgr.fetch_feature((idx, None, None))[-1]
        O.dist_fetch(ref, lists)
        assert out.shape == (len(lists[rank]), dim)
        assert out.cpu().numpy().tobytes() == feat[lists[rank]].tobytes(), f"rank {rank} step {step}: rows differ"
        hit, miss, bad = mgr.COALA_GNN_Cache.stats()
        assert (hit, miss, bad) == (ref[rank].hit_cnt, ref[rank].miss_cnt, 0), f"rank {rank} step {step}: owner counters differ"
    keys, cnt, _ = mgr.COALA_GNN_Cache.dump()
    assert np.array_equal(keys, ref[rank].keys()) and np.array_equal(cnt, ref[rank].set_cnt())
    assert ref[rank].hit_cnt > 0
    dist.barrier()
    del mgr
    shard.close()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
