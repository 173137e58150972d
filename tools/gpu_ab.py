"""A/B several builds of the engine library on the bench workload: for each .so given, run
bench.py with TPAMD_LIBRARY set and print value, ms/step and the per-kernel times.
Extra bench arguments after "--"."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
for so in args:
    env = dict(os.environ, TPAMD_LIBRARY=os.path.abspath(so))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + extra, env=env,
                       capture_output=True, text=True)
    try:
        l = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
        print("%-28s %10.0f paths/s  %.4f ms/step  solved %d  %s" % (os.path.basename(so), l["value"], l["ms_per_step"],
              l["config"]["solved_paths"], {k: v["ms"] for k, v in l["roofline"]["kernels"].items()}), flush=True)
    except Exception as e:
        print(so, "FAILED", e, r.stderr[-500:], flush=True)
