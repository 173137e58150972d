"""Timing-only ablation of the sample+LP kernel (results are wrong in the ablated builds)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
eng = importlib.import_module("x-edr-trajectory-planning_amd.engine")
syn = importlib.import_module("x-edr-trajectory-planning_amd.synthetic")
eng._SO = os.path.join(ROOT, "x-edr-trajectory-planning_amd", "csrc", os.environ["ABL_SO"])
B, D, N = 1024, 7, 2000
E = eng.Engine(0)
b = syn.make_joint_batch(B, D, N)
inp = eng.upload_joint_batch(b, "cuda:0")
out = eng.alloc_joint_outputs(B, N, D, "cuda:0")
import ctypes
# time only setup + K1 by max_solver_loops irrelevant: use profile events of the engine
E.profile_enable(True)
for _ in range(2):
    try:
        E.time_joint_paths(inp, out, N, max_solver_loops=1)
    except Exception as e:
        print("err", e)
torch.cuda.synchronize()
E.profile_reset()
for _ in range(5):
    E.time_joint_paths(inp, out, N, max_solver_loops=1)
torch.cuda.synchronize()
print(os.environ["ABL_SO"], {k: round(v[0], 4) for k, v in E.profile_summary().items()})
