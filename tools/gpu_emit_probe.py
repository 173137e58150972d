"""Is the sweep kernel bound by what it moves or by what it computes? Time it with and without the
qd/qdd outputs (no emit_range: no second read of the records, 224 B/sample fewer stores) at one
resident round (1024 paths) and at eight (8192). Run ON the GPU box."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
eng = importlib.import_module("x-edr-trajectory-planning_amd.engine")
syn = importlib.import_module("x-edr-trajectory-planning_amd.synthetic")
D, N = 7, 2000
E = eng.Engine(0)
E.profile_enable(True)
for B in (256, 1024, 8192):
    b = syn.make_joint_batch(B, D, N)
    inp = eng.upload_joint_batch(b, "cuda:0")
    for derivs in (True, False):
        out = eng.alloc_joint_outputs(B, N, D, "cuda:0", with_derivs=derivs)
        for _ in range(2):
            E.time_joint_paths(inp, out, N)
        torch.cuda.synchronize()
        E.profile_reset()
        for _ in range(6):
            E.time_joint_paths(inp, out, N)
        torch.cuda.synchronize()
        print("B=%5d qd/qdd=%-5s %s" % (B, derivs, E.profile_summary()), flush=True)
        del out
