"""Cholesky region (end of the kernel build -> first kernel of the posterior's cross-kernel build) of EVERY step in a kernel trace."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
builds = [i for i, r in enumerate(rows) if "k_build_mfma" in r["Kernel_Name"]]
# big symmetric builds (grid large) start a step; the small rectangular one (cross kernel) ends the fit
big = [i for i in builds if int(r_["Grid_Size_X"] if (r_ := rows[i]).get("Grid_Size_X") else r_["Grid_Size"]) > 10000000]
for n, i in enumerate(big):
    nxt = [j for j in builds if j > i]
    if not nxt: break
    j = nxt[0]
    t0, t1 = rows[i]["e"], rows[j]["s"]
    seg = rows[i + 1:j]
    byq = {}
    for r in seg:
        byq.setdefault(r["Queue_Id"], [0, 0.0, None, None])
        q = byq[r["Queue_Id"]]; q[0] += 1; q[1] += (r["e"] - r["s"]) / 1e6
        q[2] = r["s"] if q[2] is None else q[2]; q[3] = r["e"]
    print("step", n, "fit region ms %.2f" % ((t1 - t0) / 1e6), {k: (v[0], round(v[1], 2), round((v[2] - t0) / 1e6, 2), round((v[3] - t0) / 1e6, 2)) for k, v in sorted(byq.items())})
