"""Build and run tools/plan_bench.cc on the GPU box: planners/s of the device-resident planner set,
of PlanBatch with host-side state, and of the CPU oracle.  python tools/gpu_plan_bench.py [B D threads batch]"""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "x-edr-trajectory-planning_amd"
importlib.import_module(PKG + ".engine").build_library()
host, csrc, orc = (os.path.join(ROOT, PKG, "host"), os.path.join(ROOT, PKG, "csrc"), os.path.join(ROOT, "oracle"))
subprocess.check_call(["make", "-C", host, "-s"])
subprocess.check_call(["make", "-C", orc, "-s", "libtp_oracle.so"])
exe = os.path.join(ROOT, "tools", "plan_bench")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fopenmp", "-o", exe, exe + ".cc",
                       "-L" + host, "-ltp_host", "-L" + csrc, "-ltpamd", "-L" + orc, "-ltp_oracle", "-lm",
                       "-Wl,-rpath," + host, "-Wl,-rpath," + csrc, "-Wl,-rpath," + orc])
sys.exit(subprocess.call([exe] + sys.argv[1:]))
