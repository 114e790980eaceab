#!/usr/bin/env python3
"""edge_index.npy -> csc_indptr.npy / csc_indices.npy / csc_edge_ids.npy, where the reference's loaders look for them.

Stands where examples/create_csc_graph.py:255-304 stands in the reference (which needs DGL): same --data / --path /
--dataset_size arguments, same input and output locations, conversion by COALA_GNN.datasets.csc_from_edge_index on the GPU
(or on the CPU with --device cpu).

  python tools/create_csc_graph.py --data IGB --path /data/IGB/ --dataset_size medium
  python tools/create_csc_graph.py --data OGB --path /data/ogbn_papers100M/
  python tools/create_csc_graph.py --edge_index some/edge_index.npy --num_nodes 1000000 --out some/"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

IGB_NODES = {"experimental": 100000, "small": 1000000, "medium": 10000000, "large": 100000000, "full": 269346174}  # create_csc_graph.py:261-271


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", type=str, default=None, choices=["IGB", "OGB"])
    ap.add_argument("--path", type=str, default=None)
    ap.add_argument("--dataset_size", type=str, default="experimental", choices=sorted(IGB_NODES))
    ap.add_argument("--edge_index", type=str, default=None, help="an edge_index.npy of either layout ([E, 2] or [2, E]) outside the two dataset trees")
    ap.add_argument("--num_nodes", type=int, default=None)
    ap.add_argument("--out", type=str, default=None, help="output directory (default: next to the edge list, as the reference)")
    ap.add_argument("--device", type=str, default="cuda:0")
    args = ap.parse_args()
    from COALA_GNN.datasets import csc_from_edge_index, split_edge_index
    if args.edge_index:
        edge_path, n_nodes = args.edge_index, args.num_nodes
        out_dir = args.out or os.path.dirname(os.path.abspath(edge_path))
    elif args.data == "IGB":
        out_dir = os.path.join(args.path, args.dataset_size, "processed", "paper__cites__paper")
        edge_path, n_nodes = os.path.join(out_dir, "edge_index.npy"), IGB_NODES[args.dataset_size]
    elif args.data == "OGB":
        out_dir = os.path.join(args.path, "raw")
        edge_path, n_nodes = os.path.join(out_dir, "edge_index.npy"), 111059956  # create_csc_graph.py:294
    else:
        ap.error("give --data IGB|OGB with --path, or --edge_index with --num_nodes")
    out_dir = args.out or out_dir
    t0 = time.time()
    e = np.load(edge_path, mmap_mode="r")
    src, dst = split_edge_index(e)
    if n_nodes is None:
        n_nodes = int(max(src.max(), dst.max())) + 1
    indptr, indices, edge_ids = csc_from_edge_index(torch.from_numpy(np.array(src)), torch.from_numpy(np.array(dst)),
                                                    n_nodes, device=args.device)
    print(f"Indptr shape: {tuple(indptr.shape)} indicies shape:{tuple(indices.shape)} Edge id shape: {tuple(edge_ids.shape)}")  # :280
    print(f"max node: {int(indices.max()) if indices.numel() else -1}")
    os.makedirs(out_dir, exist_ok=True)
    np.save(os.path.join(out_dir, "csc_indptr.npy"), indptr.cpu().numpy())
    np.save(os.path.join(out_dir, "csc_indices.npy"), indices.cpu().numpy())
    np.save(os.path.join(out_dir, "csc_edge_ids.npy"), edge_ids.cpu().numpy())
    print(f"{indices.numel()} edges, {n_nodes} nodes -> {out_dir} in {time.time() - t0:.1f}s")


if __name__ == "__main__":
    main()
