"""A/B several builds of the engine library on the bench workload: for each .so given, run
bench.py with TPAMD_LIBRARY set and print value, ms/step and the per-kernel times."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for so in sys.argv[1:]:
    env = dict(os.environ, TPAMD_LIBRARY=os.path.abspath(so))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True)
    try:
        l = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
        print("%-28s %10.0f paths/s  %.4f ms/step  %s" % (os.path.basename(so), l["value"], l["ms_per_step"],
              {k: v["ms"] for k, v in l["roofline"]["kernels"].items()}), flush=True)
    except Exception as e:
        print(so, "FAILED", e, r.stderr[-500:], flush=True)
