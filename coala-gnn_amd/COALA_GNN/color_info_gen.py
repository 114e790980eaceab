"""Writes color.npy / topk.npy / score.npy for a CSC graph: the job of examples/color_info_gen/generate_color_data.py:11-68
(reference), on the native Graph_Coloring (coala_coloring.cpp).  Host-only, runs once per dataset."""
import os

import numpy as np

from COALA_GNN_Pybind import Graph_Coloring

__all__ = ["color_graph", "save_color_files"]


def color_graph(indptr, indices, train_nid, topk=10, seed=1):
    """-> (color int64[N], topk_color int64[C, topk], topk_affinity float64[C, topk], num_colors, num_colored_nodes)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int64)
    train_nid = np.ascontiguousarray(train_nid, dtype=np.int64)
    n = len(indptr) - 1
    tool = Graph_Coloring(n, topk=topk, seed=seed)
    tool.set_adj_csc(indptr.ctypes.data, indices.ctypes.data)
    color = np.zeros(n, dtype=np.int64)
    tool.set_color_buffer(color.ctypes.data)
    tool.cpu_color_graph_optimized(train_nid.ctypes.data, len(train_nid))
    num_colors, num_colored = tool.get_num_color(), tool.get_num_color_node()
    tk = np.zeros(num_colors * topk, dtype=np.int64)
    sc = np.zeros(num_colors * topk, dtype=np.float64)
    tool.set_topk_color_buffer(tk.ctypes.data)
    tool.set_topk_affinity_buffer(sc.ctypes.data)
    tool.cpu_calculate_color_affinity()
    return color, tk.reshape(num_colors, topk), sc.reshape(num_colors, topk), num_colors, num_colored


def save_color_files(out_path, color, topk_color, topk_affinity):
    os.makedirs(out_path, exist_ok=True)
    np.save(os.path.join(out_path, "color.npy"), color)
    np.save(os.path.join(out_path, "topk.npy"), topk_color)
    np.save(os.path.join(out_path, "score.npy"), topk_affinity)
    return tuple(os.path.join(out_path, f) for f in ("color.npy", "topk.npy", "score.npy"))
