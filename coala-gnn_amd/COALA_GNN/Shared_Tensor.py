"""Process topology + small collectives, and shared pinned-host tensors.

Mirror of COALA-GNN-Setup/COALA_GNN/Shared_Tensor.py (reference): same class names, attributes and methods
(MPI_Comm_Manager :24-112, Shared_UVA_Tensor_Manager :118-179, NumpyDataset :13-21).  mpi4py and cupy do not exist on
the MI355X image, so:
  * rank / world size come from RANK / WORLD_SIZE (torchrun) or SLURM_PROCID / SLURM_NTASKS;
  * the MPI split / allgather bootstrap (:31-52) runs over one torch.distributed world whose default group carries
    gloo for CPU tensors and RCCL ("nccl") for GPU tensors;
  * `global_comm` / `local_comm` are small shims with the few mpi4py methods user code calls
    (Barrier, Get_rank, Get_size, allgather -- examples/sbatch_ssd_gnn_train.py:153,194,300,302);
  * tensors over foreign memory are built through __cuda_array_interface__ instead of cupy.UnownedMemory.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Dataset

from COALA_GNN_Pybind import SharedUVAManager

__all__ = ["NumpyDataset", "MPI_Comm_Manager", "Shared_UVA_Tensor_Manager", "tensor_from_pointer"]


class NumpyDataset(Dataset):  # Shared_Tensor.py:13-21
    def __init__(self, np_array):
        self.data = np_array

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return self.data[idx]


def _env_int(*names, default=None):
    for n in names:
        if n in os.environ:
            return int(os.environ[n])
    return default


class _Comm:
    """The handful of mpi4py.Comm methods the reference's user code touches, over a torch.distributed group."""

    def __init__(self, group, rank, size):
        self.group, self._rank, self._size = group, rank, size

    def Get_rank(self):
        return self._rank

    def Get_size(self):
        return self._size

    def Barrier(self):
        if self._size > 1:
            dist.barrier(group=self.group)

    def allgather(self, obj):
        if self._size == 1:
            return [obj]
        out = [None] * self._size
        dist.all_gather_object(out, obj, group=self.group)
        return out


class MPI_Comm_Manager(object):
    """Shared_Tensor.py:24-112.  MPI_Comm_Manager(node): `node` is the machine ("domain") id of this process."""

    def __init__(self, node=0, backend=None, timeout_s=1800):
        self.global_rank = _env_int("RANK", "SLURM_PROCID", default=0)
        self.global_size = _env_int("WORLD_SIZE", "SLURM_NTASKS", default=1)
        self._owns_world = False
        if self.global_size > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:
                backend = "cpu:gloo,cuda:nccl" if torch.cuda.is_available() else "gloo"
            import datetime
            dist.init_process_group(backend, world_size=self.global_size, rank=self.global_rank,
                                    timeout=datetime.timedelta(seconds=timeout_s))
            self._owns_world = True
        if dist.is_initialized():
            self.global_rank, self.global_size = dist.get_rank(), dist.get_world_size()
        self.global_comm = _Comm(None, self.global_rank, self.global_size)

        self.node_id = int(node)
        # comm.Split(color=node_id, key=global_rank)  (:31): members ordered by global rank
        node_ids = self.global_comm.allgather(self.node_id)
        self.dist_local_rank_list = [r for r, nid in enumerate(node_ids) if nid == self.node_id]   # :35
        self.local_rank = self.dist_local_rank_list.index(self.global_rank)                          # :32
        self.local_size = len(self.dist_local_rank_list)                                             # :33
        self.is_master = self.local_rank == 0                                                        # :36
        # masters in global-rank order (:40-43); domains listed in the order of their masters (:45-49)
        domains = {}
        for r, nid in enumerate(node_ids):
            domains.setdefault(nid, []).append(r)
        self.master_process_list = sorted(members[0] for members in domains.values())
        self.num_master_process = len(self.master_process_list)
        by_master = {members[0]: members for members in domains.values()}
        self.global_dist_local_rank_list = [by_master[m] for m in self.master_process_list]
        self.master_process_id = self.dist_local_rank_list[0]                                        # :51-55
        self.master_process_index = self.master_process_list.index(self.master_process_id)           # :56
        # local_comm = comm.Split(...) (:31): a gloo group per domain, created by every rank in the same order
        boot = None
        if self.global_size > 1 and dist.is_initialized():
            boots = [dist.new_group(ranks=m, backend="gloo") for m in self.global_dist_local_rank_list]
            boot = boots[self.master_process_index]
        self.local_comm = _Comm(boot, self.local_rank, self.local_size)
        self.local_gloo_scatter_array = []
        self.local_gloo_gather_array = []
        self.nccl_cache_gather_array = []
        self.local_gloo_scatter = self.local_gloo_gather = self.master_gloo_gather = None
        self.nccl_cache_gather = None
        self._groups_ready = False

    # Shared_Tensor.py:62-92
    def initialize_nested_process_group(self, cache_backend="nvshmem"):
        if self._groups_ready:
            return
        self.cache_backend = cache_backend
        if self.global_size == 1 or not dist.is_initialized():
            self._groups_ready = True
            return
        use_cuda = torch.cuda.is_available()
        for members in self.global_dist_local_rank_list:  # every rank creates every group, in the same order
            self.local_gloo_scatter_array.append(dist.new_group(ranks=members, backend="gloo"))
            self.local_gloo_gather_array.append(dist.new_group(ranks=members, backend="gloo"))
            if cache_backend in ("nccl", "nvshmem"):
                # the reference builds this group only for "nccl" (:76-79); here "nvshmem" rides RCCL too
                cache_be = os.environ.get("COALA_CACHE_GROUP_BACKEND", "nccl" if use_cuda else "gloo")
                self.nccl_cache_gather_array.append(dist.new_group(ranks=members, backend=cache_be))
        self.local_gloo_gather = self.local_gloo_gather_array[self.master_process_index]
        self.local_gloo_scatter = self.local_gloo_scatter_array[self.master_process_index]
        if self.nccl_cache_gather_array:
            self.nccl_cache_gather = self.nccl_cache_gather_array[self.master_process_index]
        self.master_gloo_gather = dist.new_group(ranks=self.master_process_list, backend="gloo")
        dist.barrier()
        self._groups_ready = True

    def gather_cache_meta(self, gpu_cache_meta, gathered_data):  # Shared_Tensor.py:95-100
        if self.local_size > 1:
            dist.all_reduce(gpu_cache_meta, op=dist.ReduceOp.SUM, group=self.local_gloo_gather)
            dist.barrier(group=self.local_gloo_gather)
        if self.is_master:
            if self.num_master_process > 1:
                dist.all_gather(gathered_data, gpu_cache_meta, group=self.master_gloo_gather)
            else:
                gathered_data[0].copy_(gpu_cache_meta)

    def broadcast_training_nodes(self, parsed_training_node_list):  # Shared_Tensor.py:102-103
        if self.local_size > 1:
            dist.broadcast(parsed_training_node_list, src=self.master_process_id, group=self.local_gloo_scatter)

    def destroy_process_group(self):  # Shared_Tensor.py:105-112
        if not dist.is_initialized():
            return
        dist.barrier()
        if self._owns_world:
            dist.destroy_process_group()


class _CudaArrayHolder:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 3, "strides": None}


def tensor_from_pointer(ptr, shape, dtype, device):
    """A torch tensor on `device` aliasing `ptr` (device-visible memory we do not own; keep the owner alive)."""
    npdt = np.dtype(dtype) if not isinstance(dtype, torch.dtype) else np.dtype(
        torch.empty(0, dtype=dtype).numpy().dtype)
    return torch.as_tensor(_CudaArrayHolder(ptr, shape, npdt.str), device=device)


class MemoryOwner:
    pass


class Shared_UVA_Tensor_Manager(object):
    """Shared_Tensor.py:118-179.  A POSIX shm region pinned by every local rank; get_tensor() returns the GPU alias."""

    def __init__(self, comm_manager, path, tensor_size: int):
        self.comm_manager = comm_manager
        # the GPU this rank drives: its local rank (:129), unless the topology object pins another ordinal (several ranks on one GPU)
        dev = int(getattr(comm_manager, "device_index", comm_manager.local_rank))
        self.memory_handle = SharedUVAManager(path, int(tensor_size), comm_manager.node_id, 0, 0,
                                              local_rank=comm_manager.local_rank, device=dev, barrier=comm_manager.local_comm.Barrier)
        self.tensor_size = int(tensor_size)
        self.device_ptr = self.memory_handle.get_device_ptr()
        self.host_ptr = self.memory_handle.get_host_ptr()
        self.device = "cuda:" + str(dev)
        self.owner = MemoryOwner()

    def get_tensor(self, dtype, device, tensor_shape):  # :141-150
        uva_tensor = tensor_from_pointer(self.device_ptr, tensor_shape, dtype, self.device)
        uva_tensor._coala_owner = self  # keep the mapping alive as long as the alias
        return uva_tensor

    def get_host_array(self, dtype, tensor_shape):
        """numpy view of the same bytes through the host mapping (no GPU involved)."""
        import ctypes
        n = int(np.prod(tensor_shape))
        buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(self.host_ptr)
        return np.frombuffer(buf, dtype=np.dtype(dtype)).reshape(tensor_shape)

    def write_np_array(self, uva_tensor, np_array):  # :152-162
        if self.comm_manager.local_rank == 0:
            if tuple(uva_tensor.shape) != tuple(np_array.shape):
                raise ValueError(f"Tensor shape {uva_tensor.shape} does not match numpy array shape {np_array.shape}")
            load_start = time.time()
            host = self.get_host_array(np_array.dtype, np_array.shape)
            np.copyto(host, np_array)  # plain host memcpy into the shared mapping (no PCIe round trip)
            print(f"Data loading time: {time.time() - load_start}")
        self.comm_manager.local_comm.Barrier()

    def write_np_array_gpu(self, uva_tensor, np_array, device):  # :164-179
        if self.comm_manager.local_rank == 0:
            if tuple(uva_tensor.shape) != tuple(np_array.shape):
                raise ValueError(f"Tensor shape {uva_tensor.shape} does not match numpy array shape {np_array.shape}")
            dataloader = DataLoader(NumpyDataset(np_array), batch_size=int(1024 * 1024), shuffle=False, drop_last=False)
            offset = 0
            for batch in dataloader:
                uva_tensor[offset: offset + batch.size(0)] = batch.to(device)
                offset += batch.size(0)
        self.comm_manager.local_comm.Barrier()

    def cleanup(self):
        self.memory_handle.cleanup()
