"""Summarise the Cholesky region of a rocprofv3 --kernel-trace CSV: per-queue busy time and the panel chain."""
import csv, sys, collections, json
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# last factorisation: from the kernel build (or k_factor_input, when the factor input is not fused into the build) before
# the last k_set_identity_blocks (start of the 1024-block inverses) to that kernel
end = [i for i, r in enumerate(rows) if "k_set_identity_blocks" in r["Kernel_Name"]][-1]
fi = [i for i in range(end) if "k_factor_input" in rows[i]["Kernel_Name"] or "k_build" in rows[i]["Kernel_Name"]][-1]
t0 = rows[fi]["e"]
t1 = rows[end]["s"]
reg = [r for r in rows[fi + 1:end]]
print("cholesky region ms", (t1 - t0) / 1e6, "kernels", len(reg))
byq = collections.defaultdict(list)
for r in reg: byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    def short(nm):
        for key in ("leaf", "k_gemm_nt_h3", "k_split_rows"):
            if key in nm: return key
        return nm.split("k_gemm_nt_f32")[-1][:14]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rs:
        a = agg[short(r["Kernel_Name"])]; a[0] += 1; a[1] += (r["e"] - r["s"]) / 1e6
    print("queue", q, "n", len(rs), "busy ms", round(busy / 1e6, 2), "span ms", round((rs[-1]["e"] - rs[0]["s"]) / 1e6, 2))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("    ", k, v[0], round(v[1], 3))
leaves = [r for r in reg if "leaf" in r["Kernel_Name"]]
d = sorted((r["e"] - r["s"]) / 1e3 for r in leaves)
print("leaf us: min %.1f med %.1f p90 %.1f max %.1f" % (d[0], d[len(d) // 2], d[int(len(d) * .9)], d[-1]))
pq = leaves[0]["Queue_Id"]
prs = byq[pq]
# panel chains: groups of 8 leaves
idx = [i for i, r in enumerate(prs) if "leaf" in r["Kernel_Name"]]
out = []
for g in range(0, len(idx), 8):
    a = prs[idx[g]]; last = idx[g + 8] - 1 if g + 8 < len(idx) else len(prs) - 1
    b = prs[idx[min(g + 7, len(idx) - 1)]]
    gap_prev = (a["s"] - prs[idx[g] - 1]["e"]) / 1e3 if idx[g] > 0 else 0.0
    out.append((g // 8, round((a["s"] - t0) / 1e6, 2), round((b["e"] - a["s"]) / 1e3, 1), round(gap_prev, 1)))
print("step, start ms, chain us (first leaf start -> last leaf end), idle before us")
for o in out: print(o)
