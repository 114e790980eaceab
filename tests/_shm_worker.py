"""Two local ranks sharing ONE GPU and one POSIX shm feature table (Shared_UVA_Tensor_Manager, shared_UVA.cuh:26-115).
gloo only (RCCL cannot put two ranks on one device); each rank runs its own isolated cache over the shared cold tier."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "coala-gnn_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import oracle as O  # noqa: E402


def main():
    import torch.distributed as dist
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
    torch.cuda.set_device(0)
    from COALA_GNN import MPI_Comm_Manager, Shared_UVA_Tensor_Manager
    import COALA_GNN_Pybind as P
    comm = MPI_Comm_Manager(0)
    comm.local_rank_device = 0
    dim, rows = 256, 40000
    # every local rank maps the same segment; the device alias lives on cuda:0 for both (one-GPU box)
    mgr = Shared_UVA_Tensor_Manager.__new__(Shared_UVA_Tensor_Manager)
    mgr.comm_manager = comm
    mgr.memory_handle = P.SharedUVAManager(f"/coala_test_shm_{os.environ['MASTER_PORT']}", rows * dim * 4, 0, 0, 0,
                                           local_rank=comm.local_rank, device=0, barrier=comm.local_comm.Barrier)
    mgr.tensor_size = rows * dim * 4
    mgr.device_ptr, mgr.host_ptr = mgr.memory_handle.get_device_ptr(), mgr.memory_handle.get_host_ptr()
    mgr.device = "cuda:0"
    feat = O.make_features(rows, dim, seed=8)
    uva = mgr.get_tensor(np.float32, "cuda:0", (rows, dim))
    assert uva.is_cuda and tuple(uva.shape) == (rows, dim) and uva.data_ptr() == mgr.device_ptr
    mgr.write_np_array(uva, feat)                      # rank 0 writes through the host mapping, then local barrier
    assert np.array_equal(mgr.get_host_array(np.float32, (rows, dim))[::997], feat[::997])
    got = uva[torch.arange(0, rows, 991, device="cuda")].cpu().numpy()   # GPU reads the alias (zero-copy)
    assert np.array_equal(got, feat[::991])
    # the GPU-staged writer (Shared_Tensor.py:164-179) fills the same mapping
    feat2 = O.make_features(rows, dim, seed=9)
    mgr.write_np_array_gpu(uva, feat2, "cuda:0")
    torch.cuda.synchronize()
    comm.local_comm.Barrier()
    assert np.array_equal(mgr.get_host_array(np.float32, (rows, dim))[::499], feat2[::499])
    mgr.write_np_array(uva, feat)
    ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
    cache = P.Isolated_Cache(ctrl, None, rank, 1, 4, uva.data_ptr(), num_rows=rows)
    orc = O.OracleCache(4, dim, feat)
    rng = np.random.default_rng(rank)
    for _ in range(4):
        idx = rng.choice(rows // 4, size=3000, replace=False).astype(np.int64)
        out = torch.empty((3000, dim), dtype=torch.float32, device="cuda")
        cache.read_feature(out.data_ptr(), torch.from_numpy(idx).cuda().data_ptr(), 3000)
        assert out.cpu().numpy().tobytes() == orc.read_feature(idx).tobytes()
        assert cache.stats()[:2] == (orc.hit_cnt, orc.miss_cnt)
    cache.close()
    del uva
    comm.local_comm.Barrier()
    mgr.cleanup()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
