"""CPU tests of the C-ABI library without a GPU: it loads, exports every symbol include/coala_hip.h declares, and its
host-only entry points (geometry, .npy reader, node distributor) agree with the oracle and the golden vectors.
No compute kernel is launched here."""
import ctypes as C
import io
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_header_symbols_are_exported_and_bound(hiplib):
    from COALA_GNN_Pybind import _capi
    hdr = open(os.path.join(ROOT, "include", "coala_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(coala_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = C.CDLL(_capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/coala_hip.h but not exported by libcoala_hip.so"
    assert declared == set(_capi.SYMBOLS), f"ctypes table out of sync: {declared ^ set(_capi.SYMBOLS)}"
    assert _capi.load().coala_abi_version() == 4


def test_geometry_matches_reference_rules(hiplib, oracle):
    from COALA_GNN_Pybind import _capi
    L = _capi.load()
    for d in (1, 100, 128, 129, 256, 257, 512, 513, 1024):
        assert L.coala_cache_dim(d) == oracle.cache_dim(d)
    assert L.coala_cache_dim(1025) < 0 and "8KB" in _capi.last_error()   # ssd_gnn_cache.cuh:44
    for mb, cd in ((4096, 1024), (16384, 128), (16384, 1024), (1, 128), (3, 512)):
        assert L.coala_cache_num_sets(mb, cd) == oracle.num_sets(mb, cd)
    with pytest.raises(RuntimeError, match="8KB"):
        hiplib.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, 2048, True)
    c = hiplib.SSD_GNN_SSD_Controllers(1, 999, 1024, 0, 0, 100, True)
    assert (c.cache_dim, c.page_size) == (128, 512)                       # ssd_gnn_cache.cuh:34-35,47


def test_compute_fails_loudly_without_a_gpu(hiplib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ctrl = hiplib.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, 128, True)
    with pytest.raises(RuntimeError, match="libcoala_hip"):
        hiplib.Isolated_Cache(ctrl, None, 0, 1, 1, 4096, num_rows=10)     # no CPU fallback exists
    with pytest.raises(RuntimeError, match="sim_buf is 0"):
        hiplib.Isolated_Cache(ctrl, None, 0, 1, 1, 0, num_rows=10)        # NVMe tier is out of scope


def _npy_bytes(arr, version):
    buf = io.BytesIO()
    np.lib.format.write_array(buf, arr, version=version)
    return buf.getvalue()


@pytest.mark.parametrize("version", [(1, 0), (2, 0)])
def test_npy_parser_kats(hiplib, oracle, version):
    """node_distributor_pybind.cuh:37-109: v1/v2 headers, 1-D and 2-D shapes, '<i8' and '<f8'."""
    from COALA_GNN_Pybind import _capi
    L = _capi.load()

    def parse(buf, want):
        shape = (C.c_int64 * 2)()
        nd, off, descr = C.c_int(0), C.c_size_t(0), C.create_string_buffer(16)
        rc = L.coala_npy_parse(buf, len(buf), want, shape, C.byref(nd), C.byref(off), descr, 16)
        return rc, tuple(shape[i] for i in range(nd.value)), off.value, descr.value.decode()

    a1 = np.arange(17, dtype=np.int64)
    a2 = np.arange(60, dtype=np.int64).reshape(6, 10)
    f2 = np.linspace(0, 1, 60).reshape(6, 10)
    for arr, want, exp_shape, exp_descr in ((a1, 1, (17,), "<i8"), (a2, 2, (6, 10), "<i8"), (f2, 2, (6, 10), "<f8"),
                                            (f2, 1, (), "<f8"),     # the reference's score.npy case: 2-D read with the 1-D regex
                                            (a1, 2, (), "<i8")):
        buf = _npy_bytes(arr, version)
        got = parse(buf, want)
        assert got == (0, exp_shape, len(buf) - arr.nbytes, exp_descr)
        assert got == oracle.npy_parse(buf, want)
        if exp_shape:
            payload = np.frombuffer(buf, dtype=arr.dtype, offset=got[2]).reshape(exp_shape)
            assert np.array_equal(payload, arr)
    assert parse(b"NOTNUMPY" + bytes(20), 1)[0] == _capi.EFORMAT and "Not a valid .npy" in _capi.last_error()
    bad = bytearray(_npy_bytes(a1, version)); bad[6] = 9
    assert parse(bytes(bad), 1)[0] == _capi.EFORMAT and "Unsupported .npy file version" in _capi.last_error()
    assert parse(_npy_bytes(a1, version), 3)[0] == _capi.EINVAL


def _write_tables(tmp_path, color, topk, score):
    from _util import ColorFiles
    return ColorFiles(tmp_path, color, topk, score)


def test_distributor_matches_oracle_and_golden(hiplib, oracle, tmp_path):
    """node_distributor_pybind.cuh:150-222 through the C ABI, for 1, 2 and 4 domains."""
    gold = json.load(open(os.path.join(GOLD, "golden.json")))["distributor"]
    for case in gold:
        d = np.load(os.path.join(GOLD, case["name"] + ".npz"))
        files = _write_tables(tmp_path, d["color"], d["topk"], d["score"])
        items = np.ascontiguousarray(d["items"])
        n_nodes, batch, local = case["num_nodes"], case["batch"], case["local_size"]
        meta = [np.ascontiguousarray(d[f"meta{j}"]) for j in range(n_nodes)]
        for oi, off in enumerate(case["offsets"]):
            for j in range(n_nodes):
                nd = hiplib.Node_distributor_pybind(items.ctypes.data, j, batch, local, n_nodes, files.color_file,
                                                    files.topk_file, files.score_file)
                assert nd.get_num_colors() == d["topk"].shape[0]
                out = np.zeros(batch * local, dtype=np.int64)
                nd.distribute_node_with_affinity(off, out.ctypes.data, [m.ctypes.data for m in meta])
                want = oracle.distribute_node_with_affinity(items, off, batch, local, j, n_nodes, d["color"], d["topk"], d["score"], meta)
                assert np.array_equal(out, want) and out.tolist() == case["out"][oi][j]
        # every id of the global batch lands in exactly one domain, capacity-capped
        glob = items[: batch * local * n_nodes]
        allout = np.concatenate([np.array(case["out"][0][j]) for j in range(n_nodes)])
        assert sorted(allout.tolist()) == sorted(glob.tolist())


def test_distributor_single_domain_is_contiguous_striping(hiplib, tmp_path):
    # with one machine every id goes to bucket 0 in order == "baseline" striping
    from _util import synth_colors
    color, tk, sc = synth_colors(500, 6, seed=3)
    files = _write_tables(tmp_path, color, tk, sc)
    items = np.random.default_rng(0).permutation(500).astype(np.int64)
    nd = hiplib.Node_distributor_pybind(items.ctypes.data, 0, 8, 4, 1, files.color_file, files.topk_file, files.score_file)
    out = np.zeros(32, dtype=np.int64)
    meta = np.zeros(7, dtype=np.int32)
    nd.distribute_node_with_affinity(64, out.ctypes.data, [meta.ctypes.data])
    assert np.array_equal(out, items[64:96])


def test_distributor_errors(hiplib, tmp_path):
    items = np.arange(10, dtype=np.int64)
    with pytest.raises(RuntimeError, match="Unable to open file"):
        hiplib.Node_distributor_pybind(items.ctypes.data, 0, 2, 1, 1, "/nonexistent/c.npy", "/nonexistent/t.npy", "/nonexistent/s.npy")
    plain = hiplib.Node_distributor_pybind(items.ctypes.data, 1)
    out = np.zeros(2, dtype=np.int64)
    with pytest.raises(RuntimeError, match="not created with color information"):
        plain.distribute_node_with_affinity(0, out.ctypes.data, [0])
    np.save(tmp_path / "c32.npy", np.arange(10, dtype=np.int32))
    np.save(tmp_path / "t.npy", np.zeros((3, 2), dtype=np.int64))
    np.save(tmp_path / "s.npy", np.zeros((3, 2)))
    with pytest.raises(RuntimeError, match="dtype"):
        hiplib.Node_distributor_pybind(items.ctypes.data, 0, 2, 1, 1, str(tmp_path / "c32.npy"), str(tmp_path / "t.npy"), str(tmp_path / "s.npy"))


def test_distributor_rejects_out_of_range_topk(hiplib, tmp_path):
    """neighbour colours from topk.npy index the per-domain counter arrays (num_colors + 1 entries); a value outside
    [0, num_colors] must be refused when the files are loaded, not read out of bounds on every step."""
    items = np.arange(10, dtype=np.int64)
    np.save(tmp_path / "c.npy", np.ones(10, dtype=np.int64))
    np.save(tmp_path / "s.npy", np.zeros((3, 2)))
    for bad in (4, -1):
        tk = np.ones((3, 2), dtype=np.int64)
        tk[2, 1] = bad
        np.save(tmp_path / "t.npy", tk)
        with pytest.raises(RuntimeError, match="outside"):
            hiplib.Node_distributor_pybind(items.ctypes.data, 0, 2, 1, 1, str(tmp_path / "c.npy"), str(tmp_path / "t.npy"), str(tmp_path / "s.npy"))
    tk = np.full((3, 2), 3, dtype=np.int64)                  # the largest legal colour
    np.save(tmp_path / "t.npy", tk)
    hiplib.Node_distributor_pybind(items.ctypes.data, 0, 2, 1, 1, str(tmp_path / "c.npy"), str(tmp_path / "t.npy"), str(tmp_path / "s.npy"))
