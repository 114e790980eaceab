"""On-disk formats of the reference's shared-CSC datasets: csc_indptr.npy / csc_indices.npy /
csc_edge_ids.npy, node_feat.npy, node_label_*.npy, 60/20/20 masks -- examples/ssd_gnn_dataloader.py:401-563 (IGB),
:687-854 (OGB).  The feature table goes to a pinned cold tier (shared POSIX shm across the local ranks, or private), the
CSC arrays to HBM.  No dataset ships with the GPU box; tests write small .npy files in the same layout."""
import os

import numpy as np
import torch

from .Shared_Tensor import Shared_UVA_Tensor_Manager
from .sampler import CSCGraph

__all__ = ["SharedCSCDataset"]


class SharedCSCDataset(object):
    """dataset[0] -> CSCGraph with ndata['label'/'labels'/'train_mask'/'val_mask'/'test_mask'];  .feat_data -> object with
    data_ptr()/shape over the pinned feature table (what COALA_GNN_DataLoader takes as sim_buf)."""

    def __init__(self, root, comm_manager, device, num_classes=19, in_memory=False, shm_name="/coala_shared_feat"):
        self.root, self.comm, self.device = root, comm_manager, device
        feat_path = os.path.join(root, "node_feat.npy")
        feat_mm = np.load(feat_path, mmap_mode=None if in_memory else "r")     # ssd_gnn_dataloader.py:418-423
        if feat_mm.dtype != np.float32 or feat_mm.ndim != 2:
            raise ValueError("node_feat.npy must be float32 [num_nodes, dim]")
        self.num_nodes, self.dim = int(feat_mm.shape[0]), int(feat_mm.shape[1])
        nbytes = self.num_nodes * self.dim * 4
        self._shm = Shared_UVA_Tensor_Manager(comm_manager, shm_name, nbytes)        # :434
        self.feat_data = self._shm.get_tensor(np.float32, device, (self.num_nodes, self.dim))   # :435
        if comm_manager.local_rank == 0:                                             # :436 (streamed in 256 MiB pieces)
            host = self._shm.get_host_array(np.float32, (self.num_nodes, self.dim))
            step = max(1, (256 << 20) // (self.dim * 4))
            for lo in range(0, self.num_nodes, step):
                host[lo: lo + step] = feat_mm[lo: lo + step]
        comm_manager.local_comm.Barrier()
        indptr = torch.from_numpy(np.load(os.path.join(root, "csc_indptr.npy")).astype(np.int64, copy=False))   # :496-515
        indices = torch.from_numpy(np.load(os.path.join(root, "csc_indices.npy")).astype(np.int64, copy=False))
        labels_path = os.path.join(root, f"node_label_{num_classes}.npy")
        labels = torch.from_numpy(np.load(labels_path).astype(np.int64)) if os.path.exists(labels_path) else torch.zeros(self.num_nodes, dtype=torch.int64)
        n_train, n_val = int(self.num_nodes * 0.6), int(self.num_nodes * 0.2)        # :550-559
        train_mask = torch.zeros(self.num_nodes, dtype=torch.bool)
        val_mask = torch.zeros(self.num_nodes, dtype=torch.bool)
        test_mask = torch.zeros(self.num_nodes, dtype=torch.bool)
        train_mask[:n_train] = True
        val_mask[n_train: n_train + n_val] = True
        test_mask[n_train + n_val:] = True
        nd = {"label": labels.to(device), "labels": labels.to(device), "train_mask": train_mask, "val_mask": val_mask,
              "test_mask": test_mask}
        self.graph = CSCGraph(indptr.to(device), indices.to(device), ndata=nd)       # :523 (HBM instead of UVA)

    def __getitem__(self, i):
        return self.graph

    def __len__(self):
        return 1

    def close(self):
        self.graph.close()
        self.feat_data = None
        self._shm.cleanup()
