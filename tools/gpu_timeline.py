"""Kernel timeline of the pipelined bench (run ON the GPU box): for each library given, one
`rocprofv3 --kernel-trace` run of bench.py and the steady-state timings per step -- how long the
sampling/LP kernel (K1) and the sweep (K2) last while they share the GPU, which of them the next
sweep waits for and for how long.   python tools/gpu_timeline.py TAG lib.so [lib2.so ...] [-- bench args]"""
import glob, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
tag, libs = args[0], args[1:]
for so in libs:
    name = os.path.basename(so)
    d = os.path.join(ROOT, "gpurun_out", tag, name)
    cmd = ["rocprofv3", "--kernel-trace", "-d", d, "-o", "kt", "--output-format", "rocpd", "--", "python3",
           os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--no-cpu-baseline",
           "--no-kernel-timing"] + extra
    env = dict(os.environ, TMPDIR="/tmp", TPAMD_LIBRARY=os.path.abspath(so))
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    if r.returncode != 0:
        print(name, "FAILED", r.stderr[-800:])
        continue
    db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
    con = sqlite3.connect(db)
    rows = [(n, s, e) for n, s, e in con.execute("select name, start, end from kernels order by start") if "tpamd" in n]
    k2 = [(s, e) for n, s, e in rows if "k_sweep" in n]
    k1 = [(s, e) for n, s, e in rows if "k_sample_lp" in n or "k_cartesian_lp" in n]
    # steady state: the last 8 sweeps
    k2 = k2[-9:]
    out = []
    for (s0, e0), (s1, e1) in zip(k2[:-1], k2[1:]):
        prev_k1 = [k for k in k1 if k[1] <= s1 + 1000 and k[1] > s0]      # K1 that ended between the two sweep starts
        k1e = max([k[1] for k in prev_k1], default=None)
        k1d = max([k[1] - k[0] for k in prev_k1], default=0)
        out.append(((s1 - s0) / 1e3, (e0 - s0) / 1e3, k1d / 1e3, (s1 - e0) / 1e3,
                    (s1 - k1e) / 1e3 if k1e else float("nan")))
    n = len(out)
    mean = [sum(o[i] for o in out) / n for i in range(5)]
    print("%-22s step %.1f us | K2 %.1f us | K1 under it %.1f us | next K2 starts %.1f us after K2, %.1f us after K1"
          % (name, *mean), flush=True)
