"""Edge list -> CSC (COALA_GNN.datasets.csc_from_edge_index; the reference does it with DGL, examples/create_csc_graph.py:274-286):
host logic on CPU tensors against a numpy restatement (stable argsort by destination)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _want(src, dst, n):
    perm = np.argsort(dst, kind="stable")
    indptr = np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=n))]).astype(np.int64)
    return indptr, src[perm], perm


@pytest.mark.parametrize("n,e,seed", [(1, 0, 0), (5, 0, 1), (7, 40, 2), (1000, 20000, 3), (50000, 200000, 4)])
def test_csc_from_edge_index_matches_stable_sort(n, e, seed):
    from COALA_GNN.datasets import csc_from_edge_index
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, size=e).astype(np.int64)          # duplicates, self loops and isolated nodes all occur
    dst = rng.integers(0, max(n // 2, 1), size=e).astype(np.int64)
    indptr, indices, eids = csc_from_edge_index(torch.from_numpy(src), torch.from_numpy(dst), n)
    w = _want(src, dst, n)
    assert indptr.dtype == indices.dtype == eids.dtype == torch.int64
    assert np.array_equal(indptr.numpy(), w[0]) and np.array_equal(indices.numpy(), w[1]) and np.array_equal(eids.numpy(), w[2])
    # the properties a CSC must have whatever the order inside a column
    assert indptr[0] == 0 and indptr[-1] == e and bool((indptr[1:] >= indptr[:-1]).all())
    assert np.array_equal(dst[eids.numpy()], np.repeat(np.arange(n), np.diff(indptr.numpy())))


def test_csc_from_edge_index_rejects_bad_input():
    from COALA_GNN.datasets import csc_from_edge_index, split_edge_index
    with pytest.raises(ValueError):
        csc_from_edge_index(torch.tensor([0, 5]), torch.tensor([1, 2]), 5)       # source id == num_nodes
    with pytest.raises(ValueError):
        csc_from_edge_index(torch.tensor([0, 1]), torch.tensor([-1, 2]), 5)
    with pytest.raises(ValueError):
        csc_from_edge_index(torch.tensor([0, 1, 2]), torch.tensor([1, 2]), 5)
    with pytest.raises(ValueError):
        split_edge_index(np.zeros((3, 3), dtype=np.int64))
    a = np.arange(12, dtype=np.int64).reshape(6, 2)
    s, d = split_edge_index(a)                   # IGB layout: rows of (src, dst)
    assert np.array_equal(s, a[:, 0]) and np.array_equal(d, a[:, 1])
    s, d = split_edge_index(a.T.copy())          # OGB layout: [2, E]
    assert np.array_equal(s, a[:, 0]) and np.array_equal(d, a[:, 1])


@pytest.mark.parametrize("layout", ["IGB", "OGB"])
def test_create_csc_graph_tool_writes_the_reference_files(tmp_path, layout):
    rng = np.random.default_rng(5)
    n, e = 100000, 30000
    edges = np.stack([rng.integers(0, n, size=e), rng.integers(0, n, size=e)], axis=1).astype(np.int64)
    if layout == "IGB":
        d = tmp_path / "experimental" / "processed" / "paper__cites__paper"
        d.mkdir(parents=True)
        np.save(d / "edge_index.npy", edges)
        cmd = ["--data", "IGB", "--path", str(tmp_path), "--dataset_size", "experimental"]
    else:
        d = tmp_path / "x"
        d.mkdir()
        np.save(d / "edge_index.npy", edges.T.copy())
        cmd = ["--edge_index", str(d / "edge_index.npy"), "--num_nodes", str(n)]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "create_csc_graph.py"), *cmd, "--device", "cpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-2000:]
    w = _want(edges[:, 0], edges[:, 1], n)
    for name, want in zip(("csc_indptr.npy", "csc_indices.npy", "csc_edge_ids.npy"), w):
        assert np.array_equal(np.load(d / name), want), name


def test_generate_color_data_tool(tmp_path):
    """tools/generate_color_data.py (the reference's examples/color_info_gen/generate_color_data.py): an OGB-style tree holding only
    the raw edge list and float labels with NaNs -> the three colour files, equal to the library call on the same CSC and the
    same training nodes (the first 60 % of the LABELLED nodes)."""
    from COALA_GNN.color_info_gen import color_graph
    rng = np.random.default_rng(7)
    n, e = 5000, 40000
    src, dst = rng.integers(0, n, size=e).astype(np.int64), rng.integers(0, n, size=e).astype(np.int64)
    labels = (np.arange(n) % 40).astype(np.float32)
    labels[rng.random(n) < 0.5] = np.nan
    raw = tmp_path / "raw"
    raw.mkdir()
    np.save(raw / "edge_index.npy", np.stack([src, dst], axis=0))
    np.save(raw / "node_label.npy", labels)
    out_dir = tmp_path / "colors"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "generate_color_data.py"), "--data", "OGB", "--path", str(tmp_path),
                          "--num_nodes", str(n), "--out_path", str(out_dir)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "num_colors:" in out.stdout, out.stdout[-1000:] + out.stderr[-2000:]
    indptr, indices, _ = _want(src, dst, n)
    known = np.where(~np.isnan(labels))[0]
    color, tk, sc, n_col, _ = color_graph(indptr, indices, known[: int(0.6 * len(known))])
    assert np.array_equal(np.load(out_dir / "color.npy"), color)
    assert np.array_equal(np.load(out_dir / "topk.npy"), tk) and np.array_equal(np.load(out_dir / "score.npy"), sc)
    assert tk.shape == (n_col, 10) and f"num_colors: {n_col}" in out.stdout


def test_igb_large_and_full_label_files_and_masks(tmp_path, monkeypatch):
    """IGB large / full keep node_label_19.npy as a HEADERLESS float32 file (the reference reads it with np.memmap,
    examples/ssd_gnn_dataloader.py:381-387), and IGB-full's 60/20/20 masks run over its labelled prefix only (:527-546)."""
    from COALA_GNN import datasets
    n = 1000
    lab = (np.arange(n) % 19).astype(np.float32)
    p = tmp_path / "node_label_19.npy"
    lab.tofile(p)                                              # no .npy header: np.load would refuse this file
    labels, tr, va, te = datasets.load_labels_and_masks(str(p), n, "IGB", "large", 19)
    assert labels.dtype == torch.int64 and np.array_equal(labels.numpy(), lab.astype(np.int64))
    assert (int(tr.sum()), int(va.sum()), int(te.sum())) == (600, 200, 200) and bool(te[-1])
    monkeypatch.setitem(datasets.IGB_FULL_LABELLED, 19, 500)   # stand-in for 227,130,858 of 269,346,174
    labels, tr, va, te = datasets.load_labels_and_masks(str(p), n, "IGB", "full", 19)
    assert (int(tr.sum()), int(va.sum()), int(te.sum())) == (300, 100, 100)
    assert bool(tr[:300].all()) and bool(va[300:400].all()) and bool(te[400:500].all()) and not bool(te[500:].any())
    np.save(tmp_path / "hdr.npy", lab.astype(np.int64))        # medium and smaller: a regular .npy, masks over every node
    labels, tr, va, te = datasets.load_labels_and_masks(str(tmp_path / "hdr.npy"), n, "IGB", "medium", 19)
    assert np.array_equal(labels.numpy(), lab.astype(np.int64)) and int(te.sum()) == 200
