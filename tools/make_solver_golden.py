"""Regression vectors for the solver, generated with the CPU oracle (oracle/tp_oracle.c).

The reference's own tests hold no numeric s(t) outputs and the reference cannot be built in
this image, so these vectors do NOT pin parity with the reference: they
freeze what the oracle produces today, so that a later change of the oracle or of the engine
shows up as a diff against committed data. Problems: the 19 scenario cases of the reference's
solver tests (tests/scenarios.py) and 8 seeded synthetic joint-spline paths for each of
(D, N) in {(7,500), (7,2000), (6,2000), (14,1000)}.

  python tools/make_solver_golden.py      -> tests/golden/solver_oracle_derived.npz (+ sha256 inside)
"""
import hashlib
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import tpo  # noqa: E402
import scenarios  # noqa: E402

JOINT_CONFIGS = ((7, 500), (7, 2000), (6, 2000), (14, 1000))
JOINT_PATHS = 8


def compute():
    syn = importlib.import_module("x-edr-trajectory-planning_amd.synthetic")
    out = {}
    for name, (A, B, lo, hi), s0, s1, sd0, _meta in scenarios.all_cases():
        n, c = A.shape
        p = tpo.Profile(n, c)
        rc = p.setup(A, B, lo, hi, s0, s1, sd0, 0.0, 0.0)
        rc = p.optimize() if rc == 0 else rc
        out["scn/%s/status" % name] = np.array([rc], dtype=np.int32)
        out["scn/%s/time" % name] = np.array(p.time, dtype=np.float64)
        out["scn/%s/sd" % name] = np.array(p.sd, dtype=np.float64)
        out["scn/%s/sdd" % name] = np.array(p.sdd, dtype=np.float64)
        out["scn/%s/lei" % name] = np.array([p.last_extremal_index], dtype=np.int32)
    for D, N in JOINT_CONFIGS:
        b = syn.make_joint_batch(JOINT_PATHS, D, N)
        r = tpo.time_joint_batch(b["knots"], b["control_points"], b["vmax"], b["amax"],
                                 b["path_start"], b["delta"], N, nthreads=4)
        key = "joint/D%d_N%d" % (D, N)
        out[key + "/status"] = r["status"].astype(np.int32)
        out[key + "/lei"] = r["last_extremal_index"].astype(np.int32)
        out[key + "/time"] = r["t"]
        out[key + "/sd"] = r["sd"]
        out[key + "/sdd"] = r["sdd"]
        out[key + "/qdd_last_joint"] = np.ascontiguousarray(r["qdd"][:, :, -1])
    return out


def digest(arrays):
    h = hashlib.sha256()
    for k in sorted(arrays):
        h.update(k.encode())
        h.update(np.ascontiguousarray(arrays[k]).tobytes())
    return h.hexdigest()


if __name__ == "__main__":
    arrays = compute()
    arrays["sha256"] = np.frombuffer(digest(arrays).encode(), dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", "solver_oracle_derived.npz")
    np.savez_compressed(path, **arrays)
    print(path, os.path.getsize(path), "bytes,", len(arrays) - 1, "arrays")
