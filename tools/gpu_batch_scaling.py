"""Throughput of the joint pipeline against the batch size (one GPU)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
eng = importlib.import_module("x-edr-trajectory-planning_amd.engine")
syn = importlib.import_module("x-edr-trajectory-planning_amd.synthetic")
D, N = 7, 2000
E = eng.Engine(0)
for B in (256, 512, 1024, 2048, 4096, 8192):
    b = syn.make_joint_batch(B, D, N)
    inp = eng.upload_joint_batch(b, "cuda:0")
    out = eng.alloc_joint_outputs(B, N, D, "cuda:0")
    for _ in range(2):
        E.time_joint_paths(inp, out, N)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 10
    for _ in range(K):
        E.time_joint_paths(inp, out, N)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K
    print("B %5d  %.3f ms  %.0f paths/s  solved %d" % (B, el * 1e3, B / el, int((out["status"] == 0).sum())), flush=True)
