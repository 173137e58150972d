"""Collect the rocprofv3 evidence for one bench.py step and summarise it (run ON the GPU box):

    python tools/profile_step.py TAG [bench args...]

Runs `rocprofv3 ... -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline <bench args>` in
separate passes, as MI355X_MICROARCH.md prescribes (counters never together with traces; FETCH_SIZE
and WRITE_SIZE do not fit one pass):
  1. --kernel-trace --stats              -> TAG_kernel_stats.csv
  2. --pmc FETCH_SIZE                    \
  3. --pmc WRITE_SIZE                    -> TAG_pmc_hbm.csv
  4. --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
  5. --pmc VALUBusy VALUUtilization      -> TAG_pmc_valu.csv
and writes gpurun_out/TAG/counters.json: per kernel the mean launch duration, HBM bytes per
launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (KiB units; gfx950 counts a 128-byte read request
as 64 bytes), VALU instructions, VALUBusy -- together with the SHA-256 of the kernel sources
they were measured on. bench.py reports roofline.traffic from profiles/counters.json only while
that hash still matches the sources. Copy the files into profiles/ to have them judged.
"""
import csv
import glob
import hashlib
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "x-edr-trajectory-planning_amd"

SHORT = (("k_sweep", "k_sweep"), ("k_sample_lp_joint", "k_sample_lp"), ("k_lp_rows", "k_sample_lp"),
         ("k_cartesian_lp", "k_sample_lp"), ("k_boundary_zfit", "k_boundary_zfit"),
         ("k_boundary_detect", "k_boundary_detect"), ("k_boundary_final", "k_boundary_final"),
         ("k_epilogue", "k_epilogue"), ("k_setup", "k_setup"))


def short(name):
    for key, s in SHORT:
        if key in name:
            return s
    return None


def source_hash():
    """SHA-256 over the kernel sources (the same function bench.py uses)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, PKG, "csrc")
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            with open(os.path.join(csrc, fn), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def run_pass(outdir, name, flags, bench_args):
    d = os.path.join(outdir, name)
    cmd = (["rocprofv3"] + flags + ["-d", d, "-o", name, "--output-format", "rocpd", "--",
                                    "python3", os.path.join(ROOT, "bench.py"), "--steps", "5",
                                    "--warmup", "2", "--no-cpu-baseline"] + bench_args)
    print("[profile_step]", " ".join(cmd), flush=True)
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    with open(os.path.join(outdir, name + ".log"), "w") as f:
        f.write(r.stdout)
        f.write(r.stderr)
    if r.returncode != 0:
        print(r.stderr[-2000:])
        raise SystemExit("rocprofv3 pass %s failed (%d)" % (name, r.returncode))
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if not dbs:
        raise SystemExit("pass %s wrote no database" % name)
    line = [l for l in r.stdout.splitlines() if l.startswith("{\"metric\"")]
    return dbs[0], (json.loads(line[-1]) if line else None)


def counter_means(db, counters):
    con = sqlite3.connect(db)
    out = {}
    q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
         "group by kernel_name, counter_name")
    for name, counter, calls, mean in con.execute(q):
        if counter in counters:
            out.setdefault(name, {})[counter] = (calls, mean)
    return out


def main():
    tag = sys.argv[1]
    bench_args = sys.argv[2:]
    outdir = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(outdir, exist_ok=True)
    kt, line = run_pass(outdir, "kt", ["--kernel-trace", "--stats"], bench_args)
    fetch, _ = run_pass(outdir, "fetch", ["--pmc", "FETCH_SIZE"], bench_args)
    write, _ = run_pass(outdir, "write", ["--pmc", "WRITE_SIZE"], bench_args)
    inst, _ = run_pass(outdir, "inst", ["--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
                                        "SQ_WAVES"], bench_args)
    busy, _ = run_pass(outdir, "busy", ["--pmc", "VALUBusy", "VALUUtilization"], bench_args)

    con = sqlite3.connect(kt)
    rows = con.execute("select name, total_calls, total_duration, average, percentage "
                       "from top_kernels").fetchall()
    with open(os.path.join(outdir, tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        w.writerows(rows)
    kernels = {}
    for name, calls, total, avg, pct in rows:
        s = short(name)
        if s and "tpamd" in name:
            k = kernels.setdefault(s, {"names": [], "avg_us": 0.0, "calls": 0})
            k["names"].append(name)
            k["avg_us"] += avg                # (us) kernels sharing a short name run once per step each
            k["calls"] = max(k["calls"], calls)

    hbm = {}
    for db, counter in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
        for name, m in counter_means(db, {counter}).items():
            hbm.setdefault(name, {}).update(m)
    with open(os.path.join(outdir, tag + "_pmc_hbm.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Launches", "FETCH_SIZE_KiB_mean", "WRITE_SIZE_KiB_mean",
                    "HBM_bytes_per_launch=(2*FETCH+WRITE)*1024"])
        for name, m in sorted(hbm.items()):
            if "tpamd" not in name:
                continue
            fs = m.get("FETCH_SIZE", (0, 0.0))
            wsz = m.get("WRITE_SIZE", (0, 0.0))
            b = int(round((2 * fs[1] + wsz[1]) * 1024))
            w.writerow([name, fs[0], round(fs[1], 2), round(wsz[1], 2), b])
            s = short(name)
            if s in kernels:
                kernels[s]["hbm_bytes"] = kernels[s].get("hbm_bytes", 0) + b
                kernels[s]["fetch_kib"] = round(kernels[s].get("fetch_kib", 0) + fs[1], 1)
                kernels[s]["write_kib"] = round(kernels[s].get("write_kib", 0) + wsz[1], 1)

    valu = {}
    for db, counters in ((inst, {"SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"}),
                         (busy, {"VALUBusy", "VALUUtilization"})):
        for name, m in counter_means(db, counters).items():
            valu.setdefault(name, {}).update(m)
    with open(os.path.join(outdir, tag + "_pmc_valu.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Launches", "Mean"])
        for name, m in sorted(valu.items()):
            if "tpamd" not in name:
                continue
            for counter, (calls, mean) in sorted(m.items()):
                w.writerow([name, counter, calls, round(mean, 3)])
            s = short(name)
            if s in kernels:
                k = kernels[s]
                for counter, key in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_INSTS_SALU", "salu_insts"),
                                     ("SQ_INSTS_LDS", "lds_insts"), ("SQ_WAVES", "waves")):
                    if counter in m:
                        k[key] = k.get(key, 0) + int(round(m[counter][1]))
                # busy percentages: keep the longest kernel's under a shared short name
                if "VALUBusy" in m and m["VALUBusy"][1] >= k.get("valu_busy_pct", -1):
                    k["valu_busy_pct"] = round(m["VALUBusy"][1], 2)
                    k["valu_util_pct"] = round(m.get("VALUUtilization", (0, 0.0))[1], 2)
    for k in kernels.values():
        k["avg_us"] = round(k["avg_us"], 2)
    cfg = (line or {}).get("config", {})
    summary = {
        "tag": tag, "source_sha256": source_hash(),
        "workload": "B%d:D%d:N%d" % (cfg.get("paths_per_gpu", 0), cfg.get("num_dofs", 0),
                                     cfg.get("num_samples", 0)),
        "bench_args": bench_args,
        "bench_line_under_rocprof": line,
        "hbm_formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch (MI355X_MICROARCH.md)",
        "kernels": kernels,
    }
    with open(os.path.join(outdir, "counters.json"), "w") as f:
        json.dump(summary, f, indent=1)
    # the pass databases (~8 MB each) have been summarised: gpurun copies gpurun_out/ back only
    # while it stays under 64 MiB
    if os.environ.get("TPAMD_KEEP_ROCPD") != "1":
        for db in glob.glob(os.path.join(outdir, "**", "*.db"), recursive=True):
            os.remove(db)
    print(json.dumps(summary["kernels"], indent=1))


if __name__ == "__main__":
    main()
