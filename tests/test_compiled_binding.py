"""The compiled pybind11 module COALA_GNN_Pybind._coala_pybind (csrc/coala_pybind.cpp): the reference's class surface
(COALA_GNN_Modules/COALA_GNN_Pybind.cu:27-79) over the C ABI.  Host-only classes run in the CPU suite against the same golden
vectors and KATs as the ctypes binding; the cache classes are checked on the GPU with the reference's own call conventions."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REFERENCE_CLASSES = {"SharedUVAManager", "SSD_GNN_SSD_Controllers", "SSD_GNN_NVSHMEM_Cache", "Isolated_Cache", "Node_distributor_pybind",
                     "NVSHMEM_Manager", "Graph_Coloring"}
REFERENCE_METHODS = {
    "SharedUVAManager": {"get_host_ptr", "get_device_ptr", "cleanup"},
    "SSD_GNN_NVSHMEM_Cache": {"send_requests", "read_feature", "get_cache_data", "print_stats"},
    "Isolated_Cache": {"read_feature", "get_cache_data", "split_node_list", "nccl_get_feature", "map_feat_data", "print_stats"},
    "Node_distributor_pybind": {"distribute_node_with_affinity", "get_num_colors"},
    "NVSHMEM_Manager": {"allocate", "free", "finalize"},
    "Graph_Coloring": {"cpu_color_graph", "cpu_color_graph_optimized", "cpu_count_nearest_color", "cpu_count_nearest_color_less_memory",
                       "cpu_calculate_color_affinity", "set_color_buffer", "set_topk_color_buffer", "set_topk_affinity_buffer", "set_adj_csc",
                       "get_num_color_node", "get_num_color"},
}


def test_compiled_module_has_the_reference_surface(hiplib):
    nat = hiplib.native
    assert nat is not None, "the compiled binding was not built (coala-gnn_amd/build.py builds it)"
    assert nat.abi_version() == 4
    assert REFERENCE_CLASSES <= set(dir(nat))
    for cls, methods in REFERENCE_METHODS.items():
        assert methods <= set(dir(getattr(nat, cls))), f"{cls} lacks {methods - set(dir(getattr(nat, cls)))}"
    # the Python package keeps the same seven names and delegates its per-step calls to the compiled module
    assert REFERENCE_CLASSES <= set(dir(hiplib))


def test_compiled_geometry_and_errors(hiplib, oracle):
    nat = hiplib.native
    for d in (1, 100, 128, 129, 256, 257, 512, 513, 1024):
        c = nat.SSD_GNN_SSD_Controllers(1, 999, 1024, 0, 0, d, True)
        assert (c.cache_dim, c.page_size) == (oracle.cache_dim(d), oracle.cache_dim(d) * 4)     # ssd_gnn_cache.cuh:34-47
    with pytest.raises(RuntimeError, match="8KB"):
        nat.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, 2048, True)                             # ssd_gnn_cache.cuh:44
    ctrl = nat.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, 128, True)
    with pytest.raises(RuntimeError, match="sim_buf is 0"):
        nat.Isolated_Cache(ctrl, None, 0, 1, 1, 0, num_rows=10)
    with pytest.raises(RuntimeError, match="Unable to open file"):
        nat.Node_distributor_pybind(np.arange(4, dtype=np.int64).ctypes.data, 0, 2, 1, 1, "/nonexistent/c.npy", "/nonexistent/t.npy", "/nonexistent/s.npy")


def test_compiled_distributor_matches_golden(hiplib, oracle, tmp_path):
    """node_distributor_pybind.cuh:150-222 through the compiled class, with the reference's positional arguments."""
    from _util import ColorFiles
    nat = hiplib.native
    gold = json.load(open(os.path.join(GOLD, "golden.json")))["distributor"]
    for case in gold:
        d = np.load(os.path.join(GOLD, case["name"] + ".npz"))
        files = ColorFiles(tmp_path, d["color"], d["topk"], d["score"])
        items = np.ascontiguousarray(d["items"])
        n_nodes, batch, local = case["num_nodes"], case["batch"], case["local_size"]
        meta = [np.ascontiguousarray(d[f"meta{j}"]) for j in range(n_nodes)]
        for oi, off in enumerate(case["offsets"]):
            for j in range(n_nodes):
                nd = nat.Node_distributor_pybind(items.ctypes.data, j, batch, local, n_nodes, files.color_file, files.topk_file, files.score_file)
                assert nd.get_num_colors() == d["topk"].shape[0]
                out = np.zeros(batch * local, dtype=np.int64)
                nd.distribute_node_with_affinity(off, out.ctypes.data, [m.ctypes.data for m in meta])
                assert out.tolist() == case["out"][oi][j]


def test_compiled_coloring_matches_reference_vectors(hiplib):
    """generate_color_data.py:11-68 driven through the compiled Graph_Coloring: colours bit-exact against vectors generated from the
    reference's own source."""
    nat = hiplib.native
    d = np.load(os.path.join(GOLD, "coloring_a.npz"))
    nc, ncn, topk, seed = (int(x) for x in d["meta"])
    indptr, indices, train = (np.ascontiguousarray(d[k], dtype=np.int64) for k in ("indptr", "indices", "train"))
    n = len(indptr) - 1
    g = nat.Graph_Coloring(n, topk=topk, seed=seed)
    g.set_adj_csc(indptr.ctypes.data, indices.ctypes.data)
    color = np.zeros(n, dtype=np.int64)
    g.set_color_buffer(color.ctypes.data)
    g.cpu_color_graph_optimized(train.ctypes.data, len(train))
    assert (g.get_num_color(), g.get_num_color_node()) == (nc, ncn) and np.array_equal(color, d["color"])
    tk = np.zeros(nc * topk, dtype=np.int64)
    sc = np.zeros(nc * topk, dtype=np.float64)
    g.set_topk_color_buffer(tk.ctypes.data)
    g.set_topk_affinity_buffer(sc.ctypes.data)
    g.cpu_calculate_color_affinity()
    assert np.array_equal(sc.reshape(nc, topk), d["topk_affinity"])


@pytest.mark.gpu
def test_compiled_caches_match_oracle(hiplib, oracle):
    """The compiled Isolated_Cache with the reference's call sequence (read_feature, split_node_list / nccl_get_feature / map_feat_data,
    get_cache_data, print_stats) against the oracle; and the Python package's own classes, whose per-step calls go through the
    compiled module, stay bit-identical."""
    import torch
    from _util import ColorFiles, PinnedTable
    nat = hiplib.native
    dim, rows, cache_mb, ncol = 256, 20000, 2, 12
    feat = oracle.make_features(rows, dim, seed=17)
    table = PinnedTable(hiplib, feat)
    ctrl = nat.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
    cache = nat.Isolated_Cache(ctrl, None, 0, 1, cache_mb, table.device_ptr, num_rows=rows)      # the reference's six positional arguments
    orc = oracle.OracleCache(cache_mb, dim, feat)
    rng = np.random.default_rng(4)
    for step in range(4):
        idx = rng.choice(rows // 2, size=2500, replace=False).astype(np.int64)
        d_idx = torch.from_numpy(idx).cuda()
        out = torch.empty((len(idx), dim), dtype=torch.float32, device="cuda")
        cache.read_feature(out.data_ptr(), d_idx.data_ptr(), len(idx))                          # synchronous on return, like the reference
        want = orc.read_feature(idx, oracle.SCHED_HITS_FIRST)
        assert out.cpu().numpy().tobytes() == want.tobytes()
        assert cache.stats()[:2] == (orc.hit_cnt, orc.miss_cnt)
    cache.print_stats()
    assert cache.stats()[:2] == (0, 0)                                                          # print_stats resets (isolated_cache.h:139-140)
    # the "nccl" helpers on one logical rank: split -> serve -> map
    G, max_sample = 1, 4096
    idx = rng.choice(rows, size=3000, replace=False).astype(np.int64)
    d_idx = torch.from_numpy(idx).cuda()
    node = torch.zeros(G * max_sample, dtype=torch.int64, device="cuda")
    mp = torch.zeros(G * max_sample, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(G, dtype=torch.int64, device="cuda")
    cache.split_node_list(d_idx.data_ptr(), len(idx), node.data_ptr(), mp.data_ptr(), cnt.data_ptr(), G, max_sample)
    assert cnt.cpu().tolist() == [3000]
    gathered = torch.zeros((3000, dim), dtype=torch.float32, device="cuda")
    cache.nccl_get_feature([node.data_ptr()], [gathered.data_ptr()], [3000], G, max_sample)
    out = torch.zeros((3000, dim), dtype=torch.float32, device="cuda")
    cache.map_feat_data(out.data_ptr(), [gathered.data_ptr()], mp.data_ptr(), [3000], G, max_sample)
    assert out.cpu().numpy().tobytes() == feat[idx].tobytes()
    cache.close()
    table.close()
