#!/usr/bin/env python3
"""Compact per-kernel timeline of the LAST bench step in a rocprofv3 --kernel-trace CSV (for offline study):
   start_us,dur_us,queue,grid,wg,name   -- relative to the step's first kernel (k_row_sqnorm of set_train)."""
import csv, re, sys
path, out = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "k_sum" in r["Kernel_Name"]]
i0 = starts[-1] if starts else 0
while i0 > 0 and rows[i0]["s"] - rows[i0 - 1]["e"] < 200000 and "k_sum" not in rows[i0 - 1]["Kernel_Name"] and i0 > starts[-1] - 8:
    i0 -= 1
t0 = rows[i0]["s"]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|nngp::|void ", "", n)
    m = re.match(r"([A-Za-z0-9_]+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]
with open(out, "w") as f:
    f.write("start_us,dur_us,queue,grid,wg,name\n")
    for r in rows[i0:]:
        f.write("%.1f,%.1f,%s,%s,%s,%s\n" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r.get("Queue_Id", ""),
                                             r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), short(r["Kernel_Name"])))
print("wrote", out, len(rows) - i0, "kernels")
