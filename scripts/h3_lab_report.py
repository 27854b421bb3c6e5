import csv, glob, json, sys, collections
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/h3lab"
plan = json.load(open("gpurun_out/h3_lab_plan.json"))
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_gemm_nt_h3" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
assert len(rows) == len(plan), (len(rows), len(plan))
agg = collections.defaultdict(list)
for p, r in zip(plan, rows):
    if p["round"] > 0:
        agg[(tuple(p["shape"]), p["cfg"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    (m, n, kk, lower) = k[0]
    fl = 2.0 * kk * (m * (m + 1) / 2 if lower else m * n)
    med = sorted(v)[len(v) // 2]
    print("%-28s %-16s min %8.1f med %8.1f us  %6.1f TF/s alg" % (k[0], k[1], min(v), med, fl / med / 1e6), ["%.0f" % x for x in v])
