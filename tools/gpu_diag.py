"""Diagnostic build run: where do the sweep kernel's cycles go? (not a timing run)
Slots: csrc/tpamd_sweep_joint.h JointSweep::diag; 0..23 backward wave, 24..47 forward wave."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
eng = importlib.import_module("x-edr-trajectory-planning_amd.engine")
syn = importlib.import_module("x-edr-trajectory-planning_amd.synthetic")
eng._SO = os.path.join(ROOT, "x-edr-trajectory-planning_amd", "csrc", os.environ.get("DIAG_SO", "libtpamd_diag.so"))
B, D, N = int(os.environ.get("DIAG_B", 1024)), int(os.environ.get("DIAG_D", 7)), int(os.environ.get("DIAG_N", 2000))
E = eng.Engine(0)
b = syn.make_joint_batch(B, D, N)
inp = eng.upload_joint_batch(b, "cuda:0")
out = eng.alloc_joint_outputs(B, N, D, "cuda:0")
for _ in range(3):
    E.time_joint_paths(inp, out, N)
torch.cuda.synchronize()
d = E.debug_diag(B).astype(np.float64)
names = ["extremal cycles (loop)", "wait for partner (loop end)", "crit search", "tail", "chain-block cycles",
         "first pair + loop", "chain blocks", "crit mismatches (must be 0)", "boundary-follow blocks",
         "first pair cycles", "scalar FindSdd steps", "loops", "tile fills", "fills w/o prefetch",
         "chain steps accepted", "whole kernel after set-up", "tail: sd/dt pass",
         "tail: time integral | lei+remaining qd/qdd", "literal crit walk (diag only)",
         "boundary: flags loaded (cum.)", "boundary: +zfit (cum.)", "boundary: +detect (cum.)", "boundary: +final (cum.)", "boundary: final re-fit cycles", "scalar-step cycles", "boundary-follow cycles", "init_carry cycles", "tile fills", "tile-fill cycles", "init_carry calls", "blocks from a guessed constraint"]
names[12] = "qd/qdd inside the loop"
names[13] = "tail: up to the sd/dt pass"
print("%-30s %38s | %38s" % ("B=%d D=%d N=%d" % (B, D, N), "backward wave (mean/min/max)", "forward wave (mean/min/max)"))
for k, n in enumerate(names):
    a, f = d[:, k], d[:, 32 + k]
    print("%-30s %12.0f %12.0f %12.0f | %12.0f %12.0f %12.0f" % (n, a.mean(), a.min(), a.max(), f.mean(), f.min(), f.max()))
# (the literal walk of the diagnostic build runs in the forward wave since the search moved there)
net = d[:, 32 + 15] - d[:, 32 + 18]
print("whole kernel minus the literal walk (forward wave): mean %.0f min %.0f max %.0f" % (net.mean(), net.min(), net.max()))

# per-path cycles of the whole kernel (the launch lasts as long as its slowest path): histogram for
# profiles/ (DIAG_HIST=path.json)
if os.environ.get("DIAG_HIST"):
    import json
    edges = np.linspace(net.min(), net.max(), 25)
    hist, _ = np.histogram(net, bins=edges)
    loops = d[:, 32 + 11]
    json.dump({"paths": B, "dofs": D, "samples": N, "cycles_per_path": {"mean": float(net.mean()), "min": float(net.min()),
               "p50": float(np.percentile(net, 50)), "p90": float(np.percentile(net, 90)), "p99": float(np.percentile(net, 99)),
               "max": float(net.max())}, "histogram_edges": [float(x) for x in edges], "histogram_counts": [int(x) for x in hist],
               "switching_point_loops": {"mean": float(loops.mean()), "min": float(loops.min()), "max": float(loops.max())},
               "correlation_cycles_vs_loops": float(np.corrcoef(net, loops)[0, 1]),
               "idle_share_of_a_one_round_launch": float(1.0 - net.mean() / net.max())},
              open(os.environ["DIAG_HIST"], "w"), indent=1)
