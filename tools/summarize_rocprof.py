"""Turn the rocprofv3 result databases of a profiling call into the committed summaries.

  python tools/summarize_rocprof.py TAG KT_DB FETCH_DB WRITE_DB

writes profiles/TAG_kernel_stats.csv (the --kernel-trace --stats table), profiles/TAG_pmc_hbm.csv
(mean FETCH_SIZE / WRITE_SIZE per launch and kernel, KiB) and refreshes profiles/hbm_traffic.json,
which bench.py reads for roofline.traffic. HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: the
counters are in KiB and on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, HBM / rocprofv3 section).
"""
import csv
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHORT = (("k_sweep", "k_sweep"), ("k_sample_lp_joint", "k_sample_lp"), ("k_lp_rows", "k_sample_lp"),
         ("k_boundary_zfit", "k_boundary_zfit"), ("k_cartesian_lp", "k_sample_lp"),
         ("k_boundary_detect", "k_boundary_detect"), ("k_boundary_final", "k_boundary_final"),
         ("k_epilogue", "k_epilogue"), ("k_setup", "k_setup"))


def short(name):
    for key, s in SHORT:
        if key in name:
            return s
    return None


def main():
    tag, kt, fetch, write = sys.argv[1:5]
    B, D, N = (int(x) for x in (sys.argv[5:8] if len(sys.argv) >= 8 else (1024, 7, 2000)))
    con = sqlite3.connect(kt)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        w.writerows(rows)
    means = {}
    for db, counter in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
        con = sqlite3.connect(db)
        q = ("select kernel_name, count(*), avg(value) from counters_collection "
             "where counter_name = ? group by kernel_name")
        for name, calls, mean in con.execute(q, (counter,)):
            means.setdefault(name, {})[counter] = (calls, mean)
    traffic = {}
    with open(os.path.join(ROOT, "profiles", tag + "_pmc_hbm.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Launches", "FETCH_SIZE_KiB_mean", "WRITE_SIZE_KiB_mean",
                    "HBM_bytes_per_launch=(2*FETCH+WRITE)*1024"])
        for name, m in sorted(means.items()):
            if "tpamd" not in name:
                continue
            fs = m.get("FETCH_SIZE", (0, 0.0))
            wsz = m.get("WRITE_SIZE", (0, 0.0))
            hbm = int(round((2 * fs[1] + wsz[1]) * 1024))
            w.writerow([name, fs[0], round(fs[1], 2), round(wsz[1], 2), hbm])
            s = short(name)
            if s:
                traffic["%s:B%d:D%d:N%d" % (s, B, D, N)] = hbm
    traffic["_note"] = ("HBM bytes per launch from rocprofv3 PMC (profiles/%s_pmc_hbm.csv): "
                        "(2*FETCH_SIZE+WRITE_SIZE)*1024" % tag)
    with open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
