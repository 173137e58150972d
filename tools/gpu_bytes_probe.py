"""How much of the pipelined step is bandwidth? The same batch (configs[1]) through the pipelined
engine (mode 1) with output arrays left out (timing only: a solve without q / qd / qdd is a valid
call, the arrays are optional): ms per step over a long run for each variant.
  python tools/gpu_bytes_probe.py [steps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
eng = importlib.import_module("x-edr-trajectory-planning_amd.engine")
syn = importlib.import_module("x-edr-trajectory-planning_amd.synthetic")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, D, N = 1024, 7, 2000
b = syn.make_joint_batch(B, D, N)
inp = eng.upload_joint_batch(b, "cuda:0")
for mode in (1, 0):
    for name, drop in (("all outputs", ()), ("no qd/qdd", ("qd", "qdd")), ("no q", ("q",)), ("no q, qd, qdd", ("q", "qd", "qdd"))):
        E = eng.Engine(0)
        E.set_pipelining(mode)
        E.reserve(B, N, 2 * D)
        outs = [eng.alloc_joint_outputs(B, N, D, "cuda:0") for _ in range(2)]
        for o in outs:
            for k in drop:
                o.pop(k)
        for k in range(300):
            E.time_joint_paths(inp, outs[k % 2], N)
        E.fence(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            E.time_joint_paths(inp, outs[k % 2], N)
        E.fence(); torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        print("pipelining %d  %-16s %.4f ms per step  %.3f M paths/s" % (mode, name, el * 1e3, B / el / 1e6), flush=True)
        E.close()
