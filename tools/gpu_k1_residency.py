"""Study build (-DTPAMD_K1_STUDY): when do the blocks of the sampling/LP kernel of step k+1 run
relative to the sweep workgroups of step k (pipelining mode 1)? Clock stamps of every K1 block and
every sweep workgroup; prints the cumulative picture.  Run ON the GPU box with TPAMD_LIBRARY
pointing at the study build."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
eng = importlib.import_module("x-edr-trajectory-planning_amd.engine")
syn = importlib.import_module("x-edr-trajectory-planning_amd.synthetic")
B, D, N = 1024, 7, 2000
b = syn.make_joint_batch(B, D, N)
inp = eng.upload_joint_batch(b, "cuda:0")
E = eng.Engine(0)
E.set_pipelining(1)
outs = [eng.alloc_joint_outputs(B, N, D, "cuda:0") for _ in range(2)]
steps = 41
for k in range(steps):
    E.time_joint_paths(inp, outs[k % 2], N)
E.fence(); torch.cuda.synchronize()
d = E.debug_diag(B).reshape(-1).view(np.uint64)
def split(a):
    a = a.reshape(-1, 2).copy()
    xcc = (a[:, 0] >> np.uint64(60)).astype(np.int64)
    a[:, 0] &= np.uint64((1 << 60) - 1)
    return a.astype(np.int64), xcc
k1, k1x = split(d[:32768])
(s0, s0x), (s1, s1x) = split(d[40960:40960 + 2048]), split(d[45056:45056 + 2048])
# the clocks of the 8 XCDs are not synchronised: everything relative to the XCD's own first sweep start
def norm(a, ax, base):
    out = a.copy()
    for x in range(16):
        m = ax == x
        if m.any():
            out[m] -= base[x]
    return out
first0 = np.array([s0[s0x == x, 0].min() if (s0x == x).any() else 0 for x in range(16)])
first1 = np.array([s1[s1x == x, 0].min() if (s1x == x).any() else 0 for x in range(16)])
later_is_0 = (first0 - first1)[first0 > 0].mean() > 0
prev, prevx, last, lastx, base = (s1, s1x, s0, s0x, first1) if later_is_0 else (s0, s0x, s1, s1x, first0)
prev, last, k1 = norm(prev, prevx, base), norm(last, lastx, base), norm(k1, k1x, base)
t0 = 0
tick = 1.0
dur = (prev[:, 1].max() - t0)
print("sweep k: workgroup starts within %d ticks, ends p10 %d p50 %d p90 %d max %d (ticks after its start)" % (
    prev[:, 0].max() - t0, *(np.percentile(prev[:, 1] - t0, q) for q in (10, 50, 90)), dur))
print("K1 of step k+1: first block starts %d, last block ends %d ticks after the sweep's start; next sweep starts at %d" % (
    k1[:, 0].min() - t0, k1[:, 1].max() - t0, last[:, 0].min() - t0))
bl = k1[:, 1] - k1[:, 0]
print("K1 block duration (ticks): mean %.0f p10 %.0f p50 %.0f p90 %.0f" % (bl.mean(), *(np.percentile(bl, q) for q in (10, 50, 90))))
edges = np.linspace(0, k1[:, 1].max() - t0, 21)
print("window end (ticks) | K1 blocks finished (cum.) | running at window end | sweep workgroups still running | mean duration of blocks started in window")
for e0, e1 in zip(edges[:-1], edges[1:]):
    fin = int((k1[:, 1] - t0 <= e1).sum())
    run = int(((k1[:, 0] - t0 <= e1) & (k1[:, 1] - t0 > e1)).sum())
    swr = int((prev[:, 1] - t0 > e1).sum())
    m = (k1[:, 0] - t0 > e0) & (k1[:, 0] - t0 <= e1)
    print("%10.0f | %6d | %5d | %5d | %8.0f" % (e1, fin, run, swr, bl[m].mean() if m.any() else 0))
