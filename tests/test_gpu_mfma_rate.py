"""The measurement aid vs_debug_mfma_rate returns a plausible matrix-core rate (GPU)."""
import ctypes as C

import pytest

from visual_slam_amd import _lib

pytestmark = pytest.mark.gpu


def test_mfma_rate_is_plausible_and_lower_on_random_operands():
    lib = _lib.load()
    zero, rnd = C.c_float(0), C.c_float(0)
    assert lib.vs_debug_mfma_rate(60, 0, C.byref(zero)) == 0
    assert lib.vs_debug_mfma_rate(60, 1, C.byref(rnd)) == 0
    # between a tenth of and slightly above the 2.5 PFLOP/s spec peak; real data never beats zeros by much
    assert 250.0 < rnd.value < 2800.0 and 250.0 < zero.value < 2800.0
    assert rnd.value < 1.1 * zero.value


def test_mfma_rate_rejects_bad_arguments():
    lib = _lib.load()
    out = C.c_float(0)
    assert lib.vs_debug_mfma_rate(0, 1, C.byref(out)) != 0
    assert lib.vs_debug_mfma_rate(60, 1, None) != 0
