"""Several devices from one process: the engine pool of the host mirror, BatchPathTiming's device
set, and libtpamd_multi.so (include/tpamd_multi.h: one host thread per device, one RCCL gather) --
a C++ test program (tests/cpp/test_host_multi.cc) on however many devices the box has. The N > 1
RCCL leg needs an 8-GPU node and stays unmeasured here; what runs is the same code with one rank."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, PKG_NAME


def _build():
    eng = importlib.import_module(PKG_NAME + ".engine")
    eng.build_library()
    eng.build_multi_library()
    host = os.path.join(ROOT, PKG_NAME, "host")
    csrc = os.path.join(ROOT, PKG_NAME, "csrc")
    subprocess.check_call(["make", "-C", host, "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libtp_oracle.so"])
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_multi")
    cmd = ["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-pthread", "-D__HIP_PLATFORM_AMD__",
           "-I/opt/rocm/include", "-o", exe, exe + ".cc", "-L" + host, "-ltp_host", "-L" + csrc,
           "-ltpamd_multi", "-ltpamd", "-L" + os.path.join(ROOT, "oracle"), "-ltp_oracle",
           "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + host, "-Wl,-rpath," + csrc,
           "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_multi_device_pieces_build_on_cpu():
    assert os.path.exists(_build())


def test_shard_bounds_of_the_c_abi_match_the_python_sharding():
    """tpamd_shard_bounds / _balanced (host arithmetic, no device) against sharding.shard_bounds /
    balanced_bounds, which bench.py and the gloo tests use."""
    eng = importlib.import_module(PKG_NAME + ".engine")
    shd = importlib.import_module(PKG_NAME + ".sharding")
    eng.build_library()
    lib = eng.load_library()
    rng = np.random.default_rng(3)
    for total, world in ((0, 3), (1, 4), (7, 8), (8, 8), (1024, 8), (65536, 8), (1000, 7), (5, 1)):
        begin = (C.c_int32 * (world + 1))()
        lib.tpamd_shard_bounds(total, world, begin)
        assert [(begin[r], begin[r + 1]) for r in range(world)] == \
            [shd.shard_bounds(total, world, r) for r in range(world)]
        costs = np.ascontiguousarray(rng.integers(500, 4001, size=total).astype(np.float64) *
                                     rng.choice([144.0, 196.0, 784.0], size=total))
        lib.tpamd_shard_bounds_balanced(total, costs.ctypes.data, world, begin)
        assert [(begin[r], begin[r + 1]) for r in range(world)] == shd.balanced_bounds(list(costs), world)
    lib.tpamd_device_count()     # callable without a device (returns 0 here)


@pytest.mark.gpu
def test_multi_device_pieces_on_gpu():
    exe = _build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(out.stdout[-4000:])
    print(out.stderr[-3000:])
    assert out.returncode == 0 and "ALL OK" in out.stdout
