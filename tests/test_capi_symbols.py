"""CPU tests: the C-ABI library loads and exports every symbol include/vslam_hip.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

from visual_slam_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "vslam_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "missing export " + s
    assert sorted(_lib.SYMBOLS) == syms
    assert b"gfx950" in lib.vs_version()


def test_default_options_are_the_references():
    lib = _lib.load()
    o = _lib.LMOptions()
    lib.vs_lm_default_options(ctypes.byref(o))
    assert o.max_num_iterations == 100          # tracking.cc:102
    assert o.initial_trust_region_radius == 1e4 and o.min_relative_decrease == 1e-3
    assert o.function_tolerance == 1e-6 and o.gradient_tolerance == 1e-10
    assert o.parameter_tolerance == 1e-8 and o.min_lm_diagonal == 1e-6


def test_argument_errors_do_not_need_a_gpu():
    lib = _lib.load()
    assert lib.vs_ba_create(None, 1, 1, ctypes.c_int64(0), ctypes.c_double(1), ctypes.c_double(1),
                            ctypes.c_double(0), ctypes.c_double(0)) == -1
    assert b"null" in lib.vs_last_error_string()
    n = ctypes.c_int32(7)
    assert lib.vs_match_two_way(None, 0, None, 0, ctypes.c_float(2), ctypes.c_float(1), None, None,
                                None, ctypes.byref(n)) == 0
    assert n.value == 0                          # empty inputs: no matches, as in the reference
