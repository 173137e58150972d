"""Print the tail of a rocprofv3 --kernel-trace CSV as a timeline: the last `count` engine kernels
with start / end relative to the first of them (microseconds), duration, hardware queue and grid.
  python tools/kernel_timeline.py <..._kernel_trace.csv> [count]"""
import csv
import re
import sys

path = sys.argv[1]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        if "tpamd" not in r["Kernel_Name"]:
            continue
        name = re.sub(r"^void |tpamd::|\(.*$", "", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r["Queue_Id"],
                     int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), r["Grid_Size_Y"],
                     r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"]))
rows.sort()
rows = rows[-count:]
t0 = rows[0][0]
print("%9s %9s %8s  q  %-34s %6s %5s %5s %7s %s" % ("start", "end", "dur", "kernel", "blocks", "gy", "vgpr", "lds", "scratch"))
for s, e, n, q, gx, gy, v, l, sc in rows:
    print("%9.1f %9.1f %8.1f %2s  %-34s %6d %5s %5s %7s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n, gx, gy, v, l, sc))
print("span %.1f us" % ((max(r[1] for r in rows) - t0) / 1e3))
