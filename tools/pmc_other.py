"""VALU / HBM counters of the kernels of tools/gpu_other_configs.py (run ON the GPU box):
python tools/pmc_other.py TAG "CTR_A CTR_B" ["CTR_C ..."]"""
import glob, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import profile_step as ps
tag, groups = sys.argv[1], [g.split() for g in sys.argv[2:]]
out = os.path.join(ps.ROOT, "gpurun_out", tag)
os.makedirs(out, exist_ok=True)
res = {}
for n, g in enumerate(groups):
    d = os.path.join(out, "g%d" % n)
    cmd = ["rocprofv3", "--pmc"] + g + ["-d", d, "-o", "g", "--output-format", "rocpd", "--", "python3",
                                        os.path.join(ps.ROOT, "tools", "gpu_other_configs.py")]
    r = subprocess.run(cmd, cwd=ps.ROOT, env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr[-1000:]); raise SystemExit(1)
    db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
    for name, m in ps.counter_means(db, set(g)).items():
        if "tpamd" in name:
            for c, (calls, mean) in m.items():
                res.setdefault(name.split("(")[0][-40:], {})[c] = (calls, round(mean, 1))
for k, v in sorted(res.items()):
    print(k, v)
