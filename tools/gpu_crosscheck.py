"""Runs the diagnostic build (-DTPAMD_DIAG) over a spread of workloads and reports how often
the sweep kernel's critical-point shortcut disagreed with the literal walk of
time_optimal_path_timing.cc:697-720 (must be zero). Run with
TPAMD_LIBRARY=<...>/libtpamd_diag.so; prints one JSON line."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

eng = importlib.import_module("x-edr-trajectory-planning_amd.engine")
syn = importlib.import_module("x-edr-trajectory-planning_amd.synthetic")
assert "diag" in os.path.basename(eng._SO), "set TPAMD_LIBRARY to the diagnostic library"
E = eng.Engine(0)
report = {"searches": 0, "mismatches": 0, "cases": []}
for kind, D, N, B in (("joint", 7, 2000, 256), ("joint", 6, 777, 64), ("joint", 14, 1000, 64),
                      ("joint", 7, 4096, 16), ("joint", 7, 65, 32), ("cartesian", 6, 1500, 64),
                      ("cartesian", 7, 500, 32)):
    out = eng.alloc_joint_outputs(B, N, D, "cuda:0")
    if kind == "joint":
        b = syn.make_joint_batch(B, D, N, first_path_index=1000)
        E.time_joint_paths(eng.upload_joint_batch(b, "cuda:0"), out, N)
    else:
        b = syn.make_cartesian_batch(B, D, N, first_path_index=1000)
        E.time_cartesian_paths(syn.upload_cartesian_batch(b, "cuda:0"), out)
    torch.cuda.synchronize()
    d = E.debug_diag(B)
    report["searches"] += int(d[:, 11].sum())
    report["mismatches"] += int(d[:, 7].sum() + d[:, 32 + 7].sum())   # both waves run the check
    report["cases"].append([kind, D, N, B, int((out["status"] == 0).sum())])
print(json.dumps(report))
