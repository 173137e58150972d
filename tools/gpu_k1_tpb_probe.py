"""K1 (sampling + LP) duration against its block size, per joint count (run ON the GPU box; the
engine reads TPAMD_K1_TPB when it is created)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
eng = importlib.import_module("x-edr-trajectory-planning_amd.engine")
syn = importlib.import_module("x-edr-trajectory-planning_amd.synthetic")
for D, B, N in ((6, 1024, 2000), (7, 1024, 2000), (8, 1024, 2000), (14, 512, 2000), (14, 512, 4000)):
    b = syn.make_joint_batch(B, D, N)
    row = []
    for tpb in (256, 128, 64):
        os.environ["TPAMD_K1_TPB"] = str(tpb)
        E = eng.Engine(0)
        inp = eng.upload_joint_batch(b, "cuda:0")
        out = eng.alloc_joint_outputs(B, N, D, "cuda:0")
        E.profile_enable(True)
        for _ in range(2):
            E.time_joint_paths(inp, out, N)
        torch.cuda.synchronize()
        E.profile_reset()
        for _ in range(5):
            E.time_joint_paths(inp, out, N)
        torch.cuda.synchronize()
        s = E.profile_summary()
        row.append("tpb %d: K1 %.4f ms (sweep %.4f)" % (tpb, s["k_sample_lp"][0], s["k_sweep"][0]))
        del E
    print("D=%d B=%d N=%d | %s" % (D, B, N, " | ".join(row)), flush=True)
