"""Shader clock under sustained load: the diagnostic build's per-path cycle counts (s_memtime)
against the kernel's wall time, for one resident round (1024 paths) and eight (8192). Run ON the
GPU box."""
import importlib, os, sys, subprocess, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
eng = importlib.import_module("x-edr-trajectory-planning_amd.engine")
syn = importlib.import_module("x-edr-trajectory-planning_amd.synthetic")
eng._SO = os.path.join(ROOT, "x-edr-trajectory-planning_amd", "csrc", "libtpamd_diag.so")
D, N = 7, 2000
E = eng.Engine(0)
E.profile_enable(True)
for B in (1024, 8192):
    b = syn.make_joint_batch(B, D, N)
    inp = eng.upload_joint_batch(b, "cuda:0")
    out = eng.alloc_joint_outputs(B, N, D, "cuda:0")
    for _ in range(2):
        E.time_joint_paths(inp, out, N)
    torch.cuda.synchronize()
    E.profile_reset()
    for _ in range(4):
        E.time_joint_paths(inp, out, N)
    torch.cuda.synchronize()
    ms = E.profile_summary()["k_sweep"][0]
    d = E.debug_diag(B).astype(np.float64)
    whole = d[:, 15] + d[:, 22]          # after set-up + boundary passes (cumulative)
    slots = 1024
    print("B=%d sweep %.3f ms; path cycles mean %.0f max %.0f; sum/slots = %.0f cycles -> >= %.2f GHz if the slots never idle"
          % (B, ms, whole.mean(), whole.max(), whole.sum() / slots, whole.sum() / slots / (ms * 1e6)), flush=True)
# clocks as the SMI sees them while the product library runs back to back
eng2 = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4000", "--no-cpu-baseline",
                         "--no-kernel-timing"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
time.sleep(6)
for _ in range(5):
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
    print("\n".join(l for l in r.stdout.splitlines() if "sclk" in l or "Power" in l or "fclk" in l or "mclk" in l), flush=True)
    time.sleep(0.3)
print(eng2.communicate()[0][-300:])
