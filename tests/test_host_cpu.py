"""CPU-only part of the C++ host mirror: online path edits (knot insertion, truncation,
extension, SwitchToWaypointPath, ProjectPointOnPath) against the reference's own tests restated
as data, and the mirror's brute-force LP against the oracle's (tests/cpp/test_host_cpu.cc)."""
import importlib
import os
import subprocess

from conftest import ROOT, PKG_NAME


def test_host_path_edits_and_brute_force_lp():
    importlib.import_module(PKG_NAME + ".engine").build_library()
    host = os.path.join(ROOT, PKG_NAME, "host")
    subprocess.check_call(["make", "-C", host, "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libtp_oracle.so"])
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_cpu")
    cmd = ["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-o", exe, exe + ".cc",
           "-L" + host, "-ltp_host", "-L" + os.path.join(ROOT, PKG_NAME, "csrc"), "-ltpamd",
           "-L" + os.path.join(ROOT, "oracle"), "-ltp_oracle", "-lm",
           "-Wl,-rpath," + host, "-Wl,-rpath," + os.path.join(ROOT, PKG_NAME, "csrc"),
           "-Wl,-rpath," + os.path.join(ROOT, "oracle")]
    subprocess.check_call(cmd)
    # the reference's corner-rounding fixtures (data) as a flat list of numbers for the C++ reader
    import json
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "spline_utils_golden.json")))
    flat = os.path.join(ROOT, "tests", "cpp", "spline_utils_cases.txt")
    with open(flat, "w") as f:
        f.write("%d\n" % len(fx["cases"]))
        for c in fx["cases"]:
            f.write("%d %r %r %d\n" % (len(c["corners"]), c["translation_radius"], c["rotation_radius"],
                                      c["num_control_points"]))
            for p in c["corners"]:
                f.write("%r %r %r %r\n" % (*p["translation"], p["angle"]))
            f.write("%d\n" % len(c["expected"]))
            for idx, p in sorted(c["expected"].items()):
                f.write("%s %r %r %r %r\n" % (idx, *p["translation"], p["angle"]))
    out = subprocess.run([exe, flat], capture_output=True, text=True, timeout=300)
    print(out.stdout[-3000:], out.stderr[-1000:])
    assert out.returncode == 0 and "ALL OK" in out.stdout
