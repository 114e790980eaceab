"""Seeded randomized sweep of the cache path against the oracle: random dims (vector and scalar paths, dim < cache_dim),
cache sizes (power-of-two and odd set counts), owner-partition degrees, batch shapes with duplicates / rejected ids / empty
batches, with and without colour tracking, whole serves and split serves (random redirect slice + row map, fills over random
range sets in random order).  Bit-exact rows, counters, tag table, cursors and colour counters after every batch."""
import importlib.util
import os
import subprocess
import sys

import pytest

import _fuzz_body

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the launch shape of the kernels is a tunable of the DEVELOPMENT library only (read at handle creation): the default for a host
# cold tier (64-row verdict tiles), one chunk per step as behind an HBM tier, an odd narrow grid with 16-row tiles, one-block grids,
# every tile streamed compacted / none / a low threshold with 32-row tiles
KNOBS = {2: {"COALA_K2_TILE_ROWS": "0"}, 3: {"COALA_K2_TILE_ROWS": "16", "COALA_K2_GRID": "3"},
         4: {"COALA_K2_TILE_ROWS": "64", "COALA_K2_GRID": "1", "COALA_K1_GRID": "5"}, 5: {"COALA_K1_WAVES": "4", "COALA_K1_PASSES": "2"},
         6: {"COALA_K2_SPARSE": "64"}, 7: {"COALA_K2_SPARSE": "0"}, 8: {"COALA_K2_SPARSE": "5", "COALA_K2_TILE_ROWS": "32", "COALA_K2_GRID": "2"},
         # rows in flight per wave of the probe+gather kernel: 16 passes (16 / 32 rows per chunk on short lines), 8 passes on a tiny grid
         9: {"COALA_K1_PASSES": "16"}, 10: {"COALA_K1_PASSES": "8", "COALA_K1_GRID": "3"},
         # the product takes the loop-free probe+gather kernel on lines of 1 KiB and more and the looping one on 512-B lines (and on batches beyond 1 M chunks):
         # the looping kernel on every line size, the loop-free one on every line size
         11: {"COALA_K1_SINGLE": "0"}, 12: {"COALA_K1_SINGLE": "1"}, 13: {"COALA_K1_SINGLE": "0", "COALA_K1_GRID": "2", "COALA_K1_WAVES": "1"},
         # the cold fill's dynamic deal (the default behind a host tier) with narrow grids and small tiles, so that small batches have many tiles per wave;
         # ... and the static deal (what a wide grid and small batches take)
         14: {"COALA_K2_TILE_ROWS": "16", "COALA_K2_GRID": "1"}, 15: {"COALA_K2_TILE_ROWS": "8", "COALA_K2_GRID": "2", "COALA_K2_SPARSE": "5"},
         16: {"COALA_K2_UNIT_TILES": "0", "COALA_K2_TILE_ROWS": "16", "COALA_K2_GRID": "2"}}


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_randomized_configs_match_oracle(hiplib, oracle, tmp_path, seed):
    _fuzz_body.run(hiplib, oracle, tmp_path, seed)


@pytest.mark.parametrize("seed", sorted(KNOBS))
def test_randomized_configs_other_launch_shapes(tmp_path, seed):
    """The same sweep on the development build with other grid / tile shapes of the two cache kernels (one process per shape: the
    knobs are read when the library is loaded and handles are created)."""
    spec = importlib.util.spec_from_file_location("coala_build", os.path.join(ROOT, "coala-gnn_amd", "build.py"))
    bm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bm)
    dev = bm.build_lib(dev=True)
    env = dict(os.environ, COALA_HIP_LIB=dev, **KNOBS[seed])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fuzz_body.py"), str(seed), str(tmp_path)], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and f"fuzz seed {seed} ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
