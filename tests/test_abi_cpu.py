"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950,
loads, exports every symbol include/tpamd.h declares, and refuses to run
without a GPU (no CPU fallback). No compute calls here."""
import ctypes
import importlib
import os
import re

import numpy as np
import pytest

from conftest import PKG_NAME, ROOT


@pytest.fixture(scope="module")
def eng():
    m = importlib.import_module(PKG_NAME + ".engine")
    m.build_library()
    return m


def test_header_symbols_are_all_exported(eng):
    hdr = open(os.path.join(ROOT, "include", "tpamd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tpamd_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 26
    assert declared == set(eng.ABI_SYMBOLS)
    lib = eng.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert lib.tpamd_version() == 200
    assert lib.tpamd_error_string(7).decode().startswith("could not connect")


def test_multi_device_header_symbols_are_all_exported(eng):
    """include/tpamd_multi.h against libtpamd_multi.so (loads RCCL; no device call)."""
    hdr = open(os.path.join(ROOT, "include", "tpamd_multi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tpamd_[a-z0-9_]+)\s*\(", hdr))
    assert declared == {"tpamd_multi_create", "tpamd_multi_destroy", "tpamd_multi_num_devices", "tpamd_multi_device",
                        "tpamd_multi_engine", "tpamd_multi_uses_rccl", "tpamd_gather_bytes_per_path",
                        "tpamd_multi_time_joint_paths_host", "tpamd_multi_time_joint_groups_host"}
    eng.load_library()
    lib = ctypes.CDLL(eng.build_multi_library())
    for name in sorted(declared):
        assert hasattr(lib, name), name
    lib.tpamd_gather_bytes_per_path.restype = ctypes.c_size_t
    assert lib.tpamd_gather_bytes_per_path(0, 2000, 7) == 32016      # compact: 16 N + 16
    assert lib.tpamd_gather_bytes_per_path(2, 2000, 7) == 160000     # full: north_star's t, sd, sdd... + q


def test_planner_set_struct_layouts_match_header(eng):
    class Cfg(ctypes.Structure):
        _fields_ = [(n, ctypes.c_int32) for n in ("a", "b", "c", "d", "e", "f", "g", "h")] + \
                   [("x", ctypes.c_double), ("y", ctypes.c_double), ("z", ctypes.c_int64)]
    assert ctypes.sizeof(Cfg) == 56                                   # tpamd_planner_set_config
    # tpamd_planner_summary: 3 x int64 + 8 x int32 (the device-side record has the same layout)
    assert 3 * 8 + 8 * 4 == 56


def test_struct_layouts_match_header(eng):
    # sizes implied by the C declarations (LP64): 6 int32 + double; 9 / 10 pointers; ...
    assert ctypes.sizeof(eng._JointBatch) == 32
    assert ctypes.sizeof(eng._JointInputs) == 80
    assert ctypes.sizeof(eng._PathOutputs) == 88
    assert ctypes.sizeof(eng._RowsBatch) == 16
    assert ctypes.sizeof(eng._RowsInputs) == 72
    assert ctypes.sizeof(eng._ResampleArgs) == 16 + 9 * 8 + 8 + 8 + 8 * 8


def test_no_gpu_means_loud_failure(eng):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(eng.TpamdError):
        eng.Engine(0)


def test_product_does_not_touch_the_oracle():
    """The shipped package must never import, link or call anything under oracle/."""
    pkg_dir = os.path.join(ROOT, PKG_NAME)
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cc", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "tp_oracle" not in text and "oracle." not in text.replace("oracle's", ""), f
                assert "from oracle" not in text and "import oracle" not in text, f


def test_synthetic_generator_is_deterministic_and_matches_spec():
    syn = importlib.import_module(PKG_NAME + ".synthetic")
    from oracle import tpo
    a = syn.make_joint_batch(5, 7, 500)
    b = syn.make_joint_batch(3, 7, 500, first_path_index=2)
    np.testing.assert_array_equal(a["control_points"][2:], b["control_points"])
    np.testing.assert_array_equal(a["knots"][2:], b["knots"])
    assert a["control_points"].shape == (5, 28, 7) and a["knots"].shape == (5, 31)
    assert np.all((a["vmax"] >= 1) & (a["vmax"] < 2)) and np.all((a["amax"] >= 2) & (a["amax"] < 4))
    assert np.all(np.abs(a["waypoints"]) <= 2)
    np.testing.assert_allclose(a["delta"] * 499, a["knots"][:, -1], rtol=1e-15)
    # first splitmix64 output for seed 0x5EEDC0DE00000000 (path 0), checked independently
    x = (0x5EEDC0DE00000000 + 0x9E3779B97F4A7C15) & (2**64 - 1)
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
    z ^= z >> 31
    assert a["waypoints"][0, 0, 0] == (z >> 11) * 2.0 ** -53 * 4.0 - 2.0
    # the host-side fit restates the same rule as the oracle's restatement of the reference
    for i in range(5):
        cps, knots = tpo.joint_fit_spline(a["waypoints"][i], 0.2)
        np.testing.assert_array_equal(cps, a["control_points"][i])
        np.testing.assert_array_equal(knots, a["knots"][i])
