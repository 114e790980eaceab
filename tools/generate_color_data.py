#!/usr/bin/env python3
"""color.npy / topk.npy / score.npy for a dataset directory: the reference's examples/color_info_gen/generate_color_data.py:11-68,
with its arguments (--data / --path / --dataset_size / --num_classes / --topk / --out_path), on the native Graph_Coloring.
Host-only (no GPU, no DGL): the CSC comes from csc_indptr.npy / csc_indices.npy, or from edge_index.npy when they are missing;
the training nodes from the loaders' split rule (examples/ssd_gnn_dataloader.py:550-559 IGB, :809-843 OGB).

  python tools/generate_color_data.py --data IGB --path /data/IGB/ --dataset_size medium --out_path /data/IGB/medium/"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--path", type=str, required=True)
    ap.add_argument("--dataset_size", type=str, default="experimental")
    ap.add_argument("--num_classes", type=int, default=19)
    ap.add_argument("--data", type=str, default="IGB", choices=["IGB", "OGB", "flat"])
    ap.add_argument("--out_path", type=str, default="./")
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--num_nodes", type=int, default=None, help="default: the length of csc_indptr.npy - 1, or the rows of node_feat.npy")
    args = ap.parse_args()
    from COALA_GNN.color_info_gen import color_graph, save_color_files
    from COALA_GNN.datasets import layout_paths, load_csc_arrays, load_labels_and_masks
    paths = layout_paths(args.path, args.data, args.dataset_size, args.num_classes)
    t0 = time.time()
    n = args.num_nodes
    if n is None:
        ip = os.path.join(paths["graph_dir"], "csc_indptr.npy")
        n = (len(np.load(ip, mmap_mode="r")) - 1) if os.path.exists(ip) else int(np.load(paths["feat"], mmap_mode="r").shape[0])
    print("number of nodes: ", n)                                                       # :18
    indptr, indices = load_csc_arrays(paths["graph_dir"], n, device="cpu")
    print(f"indptr len: {len(indptr)} indices len: {len(indices)}")                      # :27
    _, train_mask, _, _ = load_labels_and_masks(paths["label"], n, args.data)
    train_nid = torch.nonzero(train_mask, as_tuple=True)[0]                             # :35
    color, tk, sc, num_colors, num_colored = color_graph(indptr.numpy(), indices.numpy(), train_nid.numpy(), topk=args.topk)
    print(f"num_colors: {num_colors}")                                                   # :40
    print(f"num colored node: {num_colored}")                                            # :43
    print("saving Numpy Arrays")
    save_color_files(args.out_path, color, tk, sc)
    print(f"Saving is done ({time.time() - t0:.1f}s)")


if __name__ == "__main__":
    main()
