"""The XCD-weighted tile split of the persistent 256x256 kernel (visrep_debug_xcd_split: pure host arithmetic in the C-ABI library, no device
needed).  The eight XCDs of an MI355X run at their own clocks under the power limit; each gets whole rounds of tiles in proportion to its
measured speed (profiles/round5_gemm.md).  Checked here: the bounds tile the list exactly, equal speeds give equal shares, faster XCDs never get
fewer rounds than slower ones, and the predicted makespan is never worse than equal shares and within one round of the continuous optimum."""
import ctypes as C

import numpy as np
import pytest

from law_of_vision_representation_in_mllms_amd import _lib


def split(rel, grid, ntiles):
    lib = _lib.load()
    r = (C.c_float * 8)(*rel)
    b = (C.c_int * 9)()
    rc = lib.visrep_debug_xcd_split(r, grid, ntiles, b)
    return rc, list(b)


def makespan(rel, bounds, per8):
    return max(-(-(bounds[y + 1] - bounds[y]) // per8) * rel[y] for y in range(8))


@pytest.mark.parametrize("ntiles", [9216, 4608, 2304, 2308, 4616, 9232, 2048, 5391])
def test_equal_speeds_give_equal_shares(ntiles):
    rc, b = split([1.0] * 8, 256, ntiles)
    assert rc == 0 and b[0] == 0 and b[8] == ntiles
    sizes = np.diff(b)
    rounds = ntiles // 32
    assert (sizes >= 0).all() and sizes.sum() == ntiles
    assert set((sizes // 32).tolist()) <= {rounds // 8, rounds // 8 + 1}


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("ntiles", [9216, 4608, 2304, 2308, 9232])
def test_weighted_split_tiles_the_list_and_beats_equal_shares(seed, ntiles):
    rng = np.random.default_rng(seed)
    rel = (1.0 + rng.uniform(-0.06, 0.06, 8)).astype(np.float32)       # the measured spread is +-4 %
    rc, b = split(rel.tolist(), 256, ntiles)
    assert rc == 0 and b[0] == 0 and b[8] == ntiles and (np.diff(b) > 0).all()
    sizes = np.diff(b)
    order = np.argsort(rel)                                            # fastest first
    r = -(-sizes // 32)
    assert all(r[order[i]] >= r[order[i + 1]] - 1 for i in range(7))   # a faster XCD never runs two rounds fewer than a slower one
    _, eq = split([1.0] * 8, 256, ntiles)
    assert makespan(rel, b, 32) <= makespan(rel, eq, 32) + 1e-6
    ideal = (ntiles / 32) / (1.0 / rel).sum()                          # continuous optimum: every XCD finishes together
    assert makespan(rel, b, 32) <= ideal + rel.max() * 1.0001          # within one round of it


def test_measured_box_example():
    """Clocks of one fc1 launch (gpurun_out/r5o): 1.784 1.740 1.834 1.777 1.789 1.711 1.753 1.693 GHz -> time per round ~ 1 / clock."""
    clk = np.array([1.784, 1.740, 1.834, 1.777, 1.789, 1.711, 1.753, 1.693])
    rel = (1.0 / clk) / (1.0 / clk).mean()
    rc, b = split(rel.tolist(), 256, 9216)
    assert rc == 0
    r = np.diff(b) // 32
    assert r.sum() == 288 and r[2] == r.max() and r[7] == r.min() and r.max() - r.min() in (2, 3)      # 36 36 37 36 37 35 36 35
    _, eq = split([1.0] * 8, 256, 9216)
    gain = 1.0 - makespan(rel, b, 32) / makespan(rel, eq, 32)
    assert 0.015 < gain < 0.045


def test_rejects_unusable_arguments():
    assert split([1.0] * 8, 250, 9216)[0] == -1          # grid not a multiple of 8
    assert split([1.0] * 7 + [0.0], 256, 9216)[0] == -1  # an XCD without a measurement
    assert split([1.0] * 8, 256, 100)[0] == -1           # fewer tiles than blocks
