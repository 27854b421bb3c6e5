import csv, glob, json, sys, collections
plan = json.load(open("gpurun_out/h3_ab_plan.json"))
f = glob.glob("gpurun_out/h3ab/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_gemm_nt_h3" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
assert len(rows) == len(plan), (len(rows), len(plan))
agg = collections.defaultdict(list)
for p, r in zip(plan, rows):
    if p["round"] > 0: agg[(tuple(p["shape"]), p["variant"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items(): print(k, "min %.1f med %.1f" % (min(v), sorted(v)[len(v) // 2]), ["%.0f" % x for x in v])
