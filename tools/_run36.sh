cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python - > gpurun_out/r03_upper_dbg.log 2> gpurun_out/r03_upper_dbg.err <<'PY'
import importlib, sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
eng = importlib.import_module("x-edr-trajectory-planning_amd.engine")
syn = importlib.import_module("x-edr-trajectory-planning_amd.synthetic")
E = eng.Engine(0)
for B in (64, 256, 1024):
    b = syn.make_joint_batch(B, 7, 2000)
    inp = eng.upload_joint_batch(b, "cuda:0")
    out = eng.alloc_joint_outputs(B, 2000, 7, "cuda:0")
    for it in range(3):
        E.time_joint_paths(inp, out, 2000)
        torch.cuda.synchronize()
        print("B", B, "it", it, "ok", int((out["status"] == 0).sum()), flush=True)
PY
echo rc=$?; cat gpurun_out/r03_upper_dbg.log; grep -v "^\s*$" gpurun_out/r03_upper_dbg.err | grep -v "^  File" | head -20
