"""The C++ host mirror of the reference API (TimeOptimalPathProfile, TimeablePath,
TimeableJointSplinePath, PathTimingTrajectory, BatchPathTiming) on a real GPU: a C++
test program linked against libtp_host.so (product) and the oracle (checker)."""
import os
import subprocess

import pytest

from conftest import ROOT, PKG_NAME


def _build_test_binary():
    host = os.path.join(ROOT, PKG_NAME, "host")
    subprocess.check_call(["make", "-C", host, "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libtp_oracle.so"])
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_api")
    src = exe + ".cc"
    cmd = ["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-o", exe, src,
           "-L" + host, "-ltp_host", "-L" + os.path.join(ROOT, PKG_NAME, "csrc"), "-ltpamd",
           "-L" + os.path.join(ROOT, "oracle"), "-ltp_oracle", "-lm",
           "-Wl,-rpath," + host, "-Wl,-rpath," + os.path.join(ROOT, PKG_NAME, "csrc"),
           "-Wl,-rpath," + os.path.join(ROOT, "oracle")]
    subprocess.check_call(cmd)
    return exe


def test_host_api_builds_on_cpu():
    """CPU side: the mirror and its test program compile and link (no GPU call)."""
    import importlib
    importlib.import_module(PKG_NAME + ".engine").build_library()
    exe = _build_test_binary()
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_host_api_on_gpu():
    import importlib
    importlib.import_module(PKG_NAME + ".engine").build_library()
    exe = _build_test_binary()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(out.stdout[-4000:])
    print(out.stderr[-2000:])
    assert out.returncode == 0 and "ALL OK" in out.stdout
