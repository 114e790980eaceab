"""On-disk formats of the reference's shared-CSC datasets: csc_indptr.npy / csc_indices.npy /
csc_edge_ids.npy, node_feat.npy, node_label_*.npy, 60/20/20 masks -- examples/ssd_gnn_dataloader.py:401-563 (IGB),
:687-854 (OGB).  The feature table goes to a pinned cold tier (shared POSIX shm across the local ranks, or private), the
CSC arrays to HBM.  No dataset ships with the GPU box; tests write small .npy files in the same layout.

The reference makes the csc_*.npy files from the datasets' edge_index.npy with DGL (examples/create_csc_graph.py:255-304:
dgl.graph((src, dst)).formats('csc').adj_tensors('csc')); DGL does not exist here, so csc_from_edge_index does the conversion on the
GPU (one stable radix sort by destination: the whole of IGB-large's 1.2 G edges fits the 288 GB of one MI355X), tools/create_csc_graph.py
writes the three files where the reference puts them, and SharedCSCDataset falls back to edge_index.npy when they are missing."""
import os

import numpy as np
import torch

from .Shared_Tensor import Shared_UVA_Tensor_Manager
from .sampler import CSCGraph

__all__ = ["SharedCSCDataset", "csc_from_edge_index", "split_edge_index", "load_csc_arrays", "load_labels_and_masks", "layout_paths"]


def split_edge_index(edge_index):
    """edge_index.npy as the datasets ship it -> (src, dst) views: [E, 2] rows of (src, dst) for IGB (create_csc_graph.py:274),
    [2, E] for OGB (:295)."""
    if edge_index.ndim != 2 or 2 not in edge_index.shape:
        raise ValueError(f"edge_index must be [E, 2] or [2, E], got {tuple(edge_index.shape)}")
    if edge_index.shape[1] == 2:
        return edge_index[:, 0], edge_index[:, 1]
    return edge_index[0], edge_index[1]


def csc_from_edge_index(src, dst, num_nodes, device=None):
    """Edge list -> CSC by destination, what dgl.graph((src, dst)).adj_tensors('csc') gives the reference: indptr int64 [N + 1] over
    destination nodes, indices int64 [E] = the sources of each column, edge_ids int64 [E] = the position of each entry in the
    edge list.  Entries of a column keep the order of the edge list (one STABLE sort by destination; DGL documents no order
    within a column, and it is not installed here to compare: the sampler draws positions of a column, so its samples follow
    this order).  Runs wherever the tensors are (`device` moves them first): on the GPU it is one radix sort of E keys."""
    src = torch.as_tensor(src)
    dst = torch.as_tensor(dst)
    if device is not None:
        src, dst = src.to(device), dst.to(device)
    src, dst = src.to(torch.int64).contiguous().view(-1), dst.to(torch.int64).contiguous().view(-1)
    n = int(num_nodes)
    if src.numel() != dst.numel():
        raise ValueError("src and dst differ in length")
    if src.numel():
        lo = int(torch.minimum(src.min(), dst.min()))
        hi = int(torch.maximum(src.max(), dst.max()))
        if lo < 0 or hi >= n:
            raise ValueError(f"node ids span [{lo}, {hi}] but the graph has {n} nodes")
    key = dst.to(torch.int32) if n <= 0x7FFFFFFF else dst   # halves the sort's working set for every published dataset
    edge_ids = torch.sort(key, stable=True).indices
    del key
    indices = src[edge_ids]
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=dst.device)
    if dst.numel():
        torch.cumsum(torch.bincount(dst, minlength=n), 0, out=indptr[1:])
    return indptr, indices, edge_ids


def layout_paths(root, layout, dataset_size, num_classes):
    """Where the reference's loaders look for each file (examples/ssd_gnn_dataloader.py)."""
    if layout == "IGB":     # IGBDatast_Shared_UVA :401-563
        paper = os.path.join(root, dataset_size, "processed", "paper")
        cites = os.path.join(root, dataset_size, "processed", "paper__cites__paper")
        return {"feat": os.path.join(paper, "node_feat.npy"), "label": os.path.join(paper, f"node_label_{'19' if num_classes == 19 else '2K'}.npy"),
                "graph_dir": cites}
    if layout == "OGB":     # OGBDataset_Shared_UVA :687-854
        raw = os.path.join(root, "raw")
        return {"feat": os.path.join(raw, "node_feat.npy"), "label": os.path.join(raw, "node_label.npy"), "graph_dir": raw}
    if layout == "flat":    # everything in one directory (tests, small experiments)
        return {"feat": os.path.join(root, "node_feat.npy"), "label": os.path.join(root, f"node_label_{num_classes}.npy"), "graph_dir": root}
    raise ValueError("layout must be 'IGB', 'OGB' or 'flat'")


def load_csc_arrays(graph_dir, num_nodes, device=None):
    """csc_indptr.npy / csc_indices.npy of a dataset, or -- when they are missing -- its edge_index.npy converted (on `device`)."""
    if os.path.exists(os.path.join(graph_dir, "csc_indptr.npy")):
        indptr = torch.from_numpy(np.load(os.path.join(graph_dir, "csc_indptr.npy")).astype(np.int64, copy=False))   # :496-515
        indices = torch.from_numpy(np.load(os.path.join(graph_dir, "csc_indices.npy")).astype(np.int64, copy=False))
        return indptr, indices
    # no preprocessed CSC: convert the dataset's own edge list (the reference's DGL path, :288-319)
    e_src, e_dst = split_edge_index(np.load(os.path.join(graph_dir, "edge_index.npy"), mmap_mode="r"))
    indptr, indices, _ = csc_from_edge_index(torch.from_numpy(np.array(e_src)), torch.from_numpy(np.array(e_dst)), num_nodes, device=device)
    return indptr, indices


# IGB-full: only a prefix of the 269 M nodes carries labels (examples/ssd_gnn_dataloader.py:530-534)
IGB_FULL_LABELLED = {19: 227130858, 2983: 157675969}


def load_labels_and_masks(label_path, num_nodes, layout, dataset_size=None, num_classes=19):
    """-> (labels int64 [N] with -1 for OGB's unlabelled (NaN) nodes, train / val / test bool masks) by the reference's split rules:
    the first 60 / next 20 / last 20 % of the node ids (IGB, :550-559), of IGB-full's labelled prefix (:527-546), or of the labelled
    nodes (OGB, :809-843).  IGB large / full keep their 19-class labels as a HEADERLESS float32 file (np.memmap, :381-387)."""
    labelled = None
    if os.path.exists(label_path):
        if layout == "IGB" and dataset_size in ("large", "full") and num_classes == 19:
            raw = np.array(np.memmap(label_path, dtype=np.float32, mode="r", shape=(num_nodes,)))
        else:
            raw = np.load(label_path).reshape(-1)
        if np.issubdtype(raw.dtype, np.floating):
            nan = np.isnan(raw)
            if nan.any():
                labelled = torch.from_numpy(np.where(~nan)[0])
                raw = np.where(nan, -1, raw)
        labels = torch.from_numpy(raw.astype(np.int64))
    else:
        labels = torch.zeros(num_nodes, dtype=torch.int64)
    train_mask = torch.zeros(num_nodes, dtype=torch.bool)
    val_mask = torch.zeros(num_nodes, dtype=torch.bool)
    test_mask = torch.zeros(num_nodes, dtype=torch.bool)
    if layout == "OGB":
        pool = labelled if labelled is not None else torch.arange(num_nodes)
        n_train, n_val = int(0.6 * len(pool)), int(0.2 * len(pool))
        train_mask[pool[:n_train]] = True
        val_mask[pool[n_train: n_train + n_val]] = True
        test_mask[pool[n_train + n_val:]] = True
    else:
        pool_n = num_nodes
        if layout == "IGB" and dataset_size == "full":   # the masks stop at the labelled prefix (:530-546)
            pool_n = min(num_nodes, IGB_FULL_LABELLED[19 if num_classes == 19 else 2983])
        n_train, n_val = int(pool_n * 0.6), int(pool_n * 0.2)
        train_mask[:n_train] = True
        val_mask[n_train: n_train + n_val] = True
        test_mask[n_train + n_val: pool_n] = True
    return labels, train_mask, val_mask, test_mask


class SharedCSCDataset(object):
    """dataset[0] -> CSCGraph with ndata['label'/'labels'/'train_mask'/'val_mask'/'test_mask'];  .feat_data -> object with
    data_ptr()/shape over the pinned feature table (what COALA_GNN_DataLoader takes as sim_buf).

    layout = "IGB" (root/<dataset_size>/processed/paper/..., paper__cites__paper/...; masks = the first 60 / next 20 / last 20 % of
    the node ids, :550-559), "OGB" (root/raw/...; node_label.npy is float with NaN for unlabelled nodes and the 60/20/20 split runs
    over the labelled ones, :809-843) or "flat" (all files in root, IGB's split rule)."""

    def __init__(self, root, comm_manager, device, num_classes=19, in_memory=False, shm_name="/coala_shared_feat", layout="flat",
                 dataset_size="experimental", cold_tier="shm"):
        """cold_tier: where the feature table lives.
          "shm"         -- the reference's kind (:434-436): ONE POSIX shm segment per machine, mapped and hipHostRegister'ed by every
                           local rank.  Measured on MI355X (profiles/r03_cold_tier_kinds.txt): 41 GB take 5.0 s to create + register
                           (1.5 s per further mapping) and the zero-copy fill reads it 4 % slower than hipHostMalloc memory.
          "private"     -- the whole table in this rank's own hipHostMalloc memory (1.7 s per 41 GB; the runtime places it on the
                           GPU's NUMA node by itself): what an isolated cache on a single GPU should use.
          "partitioned" -- only the rows this rank OWNS (id % local_size == local_rank) in its own hipHostMalloc memory, for the
                           partitioned cache backends ("nccl" / "nvshmem"): 1/G of the memory per rank, next to the rank's own PCIe
                           link.  Pass cold_partitioned=dataset.cold_partitioned to COALA_GNN_DataLoader."""
        self.root, self.comm, self.device = root, comm_manager, device
        paths = layout_paths(root, layout, dataset_size, num_classes)
        feat_mm = np.load(paths["feat"], mmap_mode=None if in_memory else "r")     # ssd_gnn_dataloader.py:418-423
        if feat_mm.dtype != np.float32 or feat_mm.ndim != 2:
            raise ValueError("node_feat.npy must be float32 [num_nodes, dim]")
        if cold_tier not in ("shm", "private", "partitioned"):
            raise ValueError("cold_tier must be 'shm', 'private' or 'partitioned'")
        self.num_nodes, self.dim = int(feat_mm.shape[0]), int(feat_mm.shape[1])
        self.cold_tier, self.cold_partitioned = cold_tier, cold_tier == "partitioned"
        nbytes = self.num_nodes * self.dim * 4
        step = max(1, (256 << 20) // (self.dim * 4))
        self._shm = self._table = None
        if cold_tier == "shm":
            self._shm = Shared_UVA_Tensor_Manager(comm_manager, shm_name, nbytes)        # :434
            self.feat_data = self._shm.get_tensor(np.float32, device, (self.num_nodes, self.dim))   # :435
            if comm_manager.local_rank == 0:                                             # :436 (streamed in 256 MiB pieces)
                host = self._shm.get_host_array(np.float32, (self.num_nodes, self.dim))
                for lo in range(0, self.num_nodes, step):
                    host[lo: lo + step] = feat_mm[lo: lo + step]
            comm_manager.local_comm.Barrier()
        else:
            from .synthetic import PinnedFeatureTable
            G, r = (comm_manager.local_size, comm_manager.local_rank) if self.cold_partitioned else (1, 0)
            dev = int(getattr(comm_manager, "device_index", comm_manager.local_rank))
            local_rows = (self.num_nodes - r + G - 1) // G
            self._table = PinnedFeatureTable(local_rows, self.dim, dev)
            for lo in range(0, local_rows, step):                                        # local row k holds node id k*G + r
                hi = min(local_rows, lo + step)
                self._table.array[lo:hi] = feat_mm[r + lo * G: r + hi * G: G]
            self.feat_data = self._table                                                 # .data_ptr(): what sim_buf needs
        indptr, indices = load_csc_arrays(paths["graph_dir"], self.num_nodes, device=device)
        labels, train_mask, val_mask, test_mask = load_labels_and_masks(paths["label"], self.num_nodes, layout, dataset_size, num_classes)
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
        if self._shm is not None:
            self._shm.cleanup()
        if self._table is not None:
            self._table.close()
