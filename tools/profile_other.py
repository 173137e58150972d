"""rocprofv3 evidence for one of the other BASELINE.json configurations (run ON the GPU box):

    python tools/profile_other.py TAG CASE        # CASE: 2, 3 or 4 (tools/gpu_other_configs.py)

The same five passes as tools/profile_step.py (kernel trace + stats; FETCH_SIZE; WRITE_SIZE; SQ
instruction counts; VALUBusy / VALUUtilization -- counters never together with traces), per FULL
kernel name (the joint count is part of it), written to gpurun_out/TAG/:
  TAG_kernel_stats.csv, TAG_pmc_hbm.csv, TAG_pmc_valu.csv, TAG_counters.json
with the SHA-256 of the kernel sources they were measured on. Copy them into profiles/."""
import csv
import glob
import json
import os
import re
import sqlite3
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import profile_step as ps  # noqa: E402


def run_pass(outdir, name, flags, case):
    d = os.path.join(outdir, name)
    cmd = (["rocprofv3"] + flags + ["-d", d, "-o", name, "--output-format", "rocpd", "--", "python3",
                                    os.path.join(ps.ROOT, "tools", "gpu_other_configs.py"), case])
    print("[profile_other]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, cwd=ps.ROOT, env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    with open(os.path.join(outdir, name + ".log"), "w") as f:
        f.write(r.stdout)
        f.write(r.stderr)
    if r.returncode != 0:
        print(r.stderr[-2000:])
        raise SystemExit("rocprofv3 pass %s failed (%d)" % (name, r.returncode))
    line = [l for l in r.stdout.splitlines() if l.startswith("{\"case\"")]
    return glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0], (json.loads(line[-1]) if line else None)


def kname(full):
    return re.sub(r"^void |\(.*$", "", full)


def main():
    tag, case = sys.argv[1], sys.argv[2]
    outdir = os.path.join(ps.ROOT, "gpurun_out", tag)
    os.makedirs(outdir, exist_ok=True)
    kt, line = run_pass(outdir, "kt", ["--kernel-trace", "--stats"], case)
    fetch, _ = run_pass(outdir, "fetch", ["--pmc", "FETCH_SIZE"], case)
    write, _ = run_pass(outdir, "write", ["--pmc", "WRITE_SIZE"], case)
    inst, _ = run_pass(outdir, "inst", ["--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"], case)
    busy, _ = run_pass(outdir, "busy", ["--pmc", "VALUBusy", "VALUUtilization"], case)
    kernels = {}
    con = sqlite3.connect(kt)
    rows = [r for r in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
            if "tpamd" in r[0]]
    with open(os.path.join(outdir, tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        w.writerows(rows)
    for name, calls, total, avg, pct in rows:
        kernels[kname(name)] = {"calls": calls, "avg_us": round(avg, 2)}
    hbm = {}
    for db, counter in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
        for name, m in ps.counter_means(db, {counter}).items():
            hbm.setdefault(name, {}).update(m)
    with open(os.path.join(outdir, tag + "_pmc_hbm.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Launches", "FETCH_SIZE_KiB_mean", "WRITE_SIZE_KiB_mean",
                    "HBM_bytes_per_launch=(2*FETCH+WRITE)*1024"])
        for name, m in sorted(hbm.items()):
            if "tpamd" not in name:
                continue
            fs, wsz = m.get("FETCH_SIZE", (0, 0.0)), m.get("WRITE_SIZE", (0, 0.0))
            b = int(round((2 * fs[1] + wsz[1]) * 1024))
            w.writerow([name, fs[0], round(fs[1], 2), round(wsz[1], 2), b])
            kernels.setdefault(kname(name), {}).update(hbm_bytes=b, fetch_kib=round(fs[1], 1), write_kib=round(wsz[1], 1))
    valu = {}
    for db, counters in ((inst, {"SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"}),
                         (busy, {"VALUBusy", "VALUUtilization"})):
        for name, m in ps.counter_means(db, counters).items():
            valu.setdefault(name, {}).update(m)
    with open(os.path.join(outdir, tag + "_pmc_valu.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Launches", "Mean"])
        for name, m in sorted(valu.items()):
            if "tpamd" not in name:
                continue
            for counter, (calls, mean) in sorted(m.items()):
                w.writerow([name, counter, calls, round(mean, 3)])
            k = kernels.setdefault(kname(name), {})
            for counter, key in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_INSTS_SALU", "salu_insts"),
                                 ("SQ_INSTS_LDS", "lds_insts"), ("SQ_WAVES", "waves")):
                if counter in m:
                    k[key] = int(round(m[counter][1]))
            if "VALUBusy" in m:
                k["valu_busy_pct"] = round(m["VALUBusy"][1], 2)
                k["valu_util_pct"] = round(m.get("VALUUtilization", (0, 0.0))[1], 2)
    summary = {"tag": tag, "case": case, "source_sha256": ps.source_hash(), "line_under_rocprof": line,
               "hbm_formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch (MI355X_MICROARCH.md)",
               "kernels": kernels}
    with open(os.path.join(outdir, tag + "_counters.json"), "w") as f:
        json.dump(summary, f, indent=1)
    for db in glob.glob(os.path.join(outdir, "**", "*.db"), recursive=True):
        os.remove(db)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
