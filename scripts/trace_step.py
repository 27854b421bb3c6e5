"""Compact per-queue timeline of the LAST bench step in a rocprofv3 --kernel-trace CSV: consecutive launches of one kernel on one
queue are merged into a row (start ms from the step's first kernel, span, count, busy).  Usage: trace_step.py kernel_trace.csv [min_us]"""
import csv, sys, collections, re
path = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# a step starts with the row-norm kernel of set_train
starts = [i for i, r in enumerate(rows) if "k_row_sqnorm" in r["Kernel_Name"]]
i0 = starts[-2] if len(starts) >= 2 and len(rows) - starts[-1] < 50 else starts[-1]
step = rows[i0:]
# cut at the next step if any
t0 = step[0]["s"]
def short(nm):
    nm = re.sub(r"^void ", "", nm)
    nm = nm.replace("nngp::(anonymous namespace)::", "").replace("nngp::", "")
    nm = re.sub(r"\(.*", "", nm)
    return nm[:44]
byq = collections.defaultdict(list)
for r in step: byq[r["Queue_Id"]].append(r)
print("step: %d kernels, %.2f ms" % (len(step), (max(r["e"] for r in step) - t0) / 1e6))
for q in sorted(byq, key=lambda q: byq[q][0]["s"]):
    rs = byq[q]
    print("== queue %s: %d launches, busy %.2f ms" % (q, len(rs), sum(r["e"] - r["s"] for r in rs) / 1e6))
    groups = []
    for r in rs:
        nm = short(r["Kernel_Name"])
        if groups and groups[-1][0] == nm and r["s"] - groups[-1][2] < 200000:
            g = groups[-1]; g[2] = r["e"]; g[3] += 1; g[4] += r["e"] - r["s"]
        else:
            groups.append([nm, r["s"], r["e"], 1, r["e"] - r["s"]])
    for nm, s, e, n, busy in groups:
        if (e - s) / 1e3 >= min_us:
            print("  %9.3f ms  span %8.1f us  n %4d  busy %8.1f us  %s" % ((s - t0) / 1e6, (e - s) / 1e3, n, busy / 1e3, nm))
