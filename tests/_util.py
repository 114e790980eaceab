"""Shared helpers of the GPU parity tests (tests only)."""
import ctypes as C

import numpy as np


class PinnedTable:
    """fp32 [rows, dim] table in pinned host memory visible to the GPU (the cold tier), filled on the host."""

    def __init__(self, P, feat, device=0):
        from COALA_GNN_Pybind import _capi
        self._capi = _capi
        L = _capi.load()
        hp, dp = C.c_void_p(), C.c_void_p()
        _capi.check(L.coala_pinned_alloc(feat.nbytes, device, C.byref(hp), C.byref(dp)))
        self.host_ptr, self.device_ptr = hp.value, dp.value
        buf = (C.c_float * feat.size).from_address(self.host_ptr)
        self.array = np.frombuffer(buf, dtype=np.float32).reshape(feat.shape)
        self.array[...] = feat
        self.rows, self.dim = feat.shape

    def data_ptr(self):  # what COALA_GNN_Manager reads from sim_buf
        return self.device_ptr

    def close(self):
        if self.host_ptr:
            self.array = None
            self._capi.load().coala_pinned_free(self.host_ptr)
            self.host_ptr = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ColorFiles:
    """color.npy / topk.npy / score.npy in a temp dir, shaped like examples/color_info_gen/generate_color_data.py:39-64."""

    def __init__(self, tmpdir, color, topk, score):
        import os
        self.color_file = os.path.join(str(tmpdir), "color.npy")
        self.topk_file = os.path.join(str(tmpdir), "topk.npy")
        self.score_file = os.path.join(str(tmpdir), "score.npy")
        np.save(self.color_file, np.ascontiguousarray(color, dtype=np.int64))
        np.save(self.topk_file, np.ascontiguousarray(topk, dtype=np.int64))
        np.save(self.score_file, np.ascontiguousarray(score, dtype=np.float64))


def synth_colors(num_rows, num_colors, topk=10, seed=0):
    rng = np.random.default_rng(seed)
    color = rng.integers(0, num_colors + 1, size=num_rows).astype(np.int64)  # 0 = uncoloured
    tk = rng.integers(0, num_colors + 1, size=(num_colors, topk)).astype(np.int64)
    sc = rng.random((num_colors, topk))
    return color, tk, sc
