"""Diagnostic build run: where do the sweep kernel's cycles go? (not a timing run)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
eng = importlib.import_module("x-edr-trajectory-planning_amd.engine")
syn = importlib.import_module("x-edr-trajectory-planning_amd.synthetic")
eng._SO = os.path.join(ROOT, "x-edr-trajectory-planning_amd", "csrc", os.environ.get("DIAG_SO", "libtpamd_diag.so"))
B, D, N = int(os.environ.get("DIAG_B", 1024)), 7, 2000
E = eng.Engine(0)
b = syn.make_joint_batch(B, D, N)
inp = eng.upload_joint_batch(b, "cuda:0")
out = eng.alloc_joint_outputs(B, N, D, "cuda:0")
for _ in range(3):
    E.time_joint_paths(inp, out, N)
torch.cuda.synchronize()
d = E.debug_diag(B).astype(np.float64)
names = {0: "fwd extremals", 1: "bwd extremals", 2: "crit search", 3: "tail", 4: "chain cycles",
         5: "whole loop", 6: "chain blocks", 7: "crit search mismatches (must be 0)", 8: "n boundary fwd", 9: "first pair cycles",
         10: "n findsdd fwd", 11: "loops", 12: "tile fills", 13: "tile fills w/o prefetch",
         14: "chain steps fwd", 15: "chain steps bwd"}
for k, n in names.items():
    print("%-16s mean %12.0f  min %12.0f  max %12.0f" % (n, d[:, k].mean(), d[:, k].min(), d[:, k].max()))
nf = d[:, 10] + d[:, 11]
print("cycles per find_sdd step: %.0f" % (d[:, 4].sum() / nf.sum()))
nb = d[:, 8] + d[:, 9]
other = d[:, 0] + d[:, 1] - d[:, 4]
print("extremal cycles outside find_sdd per iteration: %.0f" % (other.sum() / (nf.sum() + nb.sum())))
iters = nf.sum() + nb.sum()
print("SUMMARY %s: whole loop %.0f cycles/path, %.0f iterations/path, %.0f cycles/iteration" % (os.environ.get("DIAG_SO", "diag"), d[:, 5].mean(), iters / B, d[:, 5].sum() / iters))
print("pre part per iteration: %.0f ; post part per iteration: %.0f" % (d[:, 7].sum() / (nf.sum() + nb.sum()), d[:, 6].sum() / (nf.sum() + nb.sum())))
