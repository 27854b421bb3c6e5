#!/bin/bash
# kernel trace of scripts/forest_scale.py (the reference's own run size) -> per-kernel totals of the whole run
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/trace_forest
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_forest -o t -- python3 $R/scripts/forest_scale.py > $R/gpurun_out/trace_forest.log 2>&1 || { tail -5 $R/gpurun_out/trace_forest.log; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/trace_forest/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# the 4th fit+predict repetition: between the 4th and 5th k_row_sqnorm... take the last full step = from the 4th set_train
starts = [i for i, r in enumerate(rows) if "k_sum" in r["Kernel_Name"] or "k_row_sqnorm" in r["Kernel_Name"]]
print("markers", len(starts))
import re
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|nngp::|void ", "", n); m = re.match(r"([A-Za-z0-9_]+(<[^>]*>)?)", n); return m.group(1) if m else n[:40]
# last level-1 predict: find last k_gemm_nt_f64 occurrences
t_end = rows[-1]["e"]
# print per-kernel totals for the 4th repetition region: heuristically the region from starts[6] to starts[8]
if len(starts) >= 8:
    a = rows[starts[6]]["s"]; b = a + 31_000_000  # one fit + predict of the 4th repetition (~30 ms)
    reg = [r for r in rows if a <= r["s"] < b]
    print("region ms", (reg[-1]["e"] - reg[0]["s"]) / 1e6, "kernels", len(reg))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in reg:
        x = agg[(r["Queue_Id"], short(r["Kernel_Name"]))]; x[0] += 1; x[1] += (r["e"] - r["s"]) / 1e6
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]: print("  ", k, v[0], round(v[1], 3))
    t0 = reg[0]["s"]
    with open("gpurun_out/timeline_forest.csv", "w") as fo:
        for r in reg: fo.write("%.1f,%.1f,%s,%s\n" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r["Queue_Id"], short(r["Kernel_Name"])))
PY
rm -rf gpurun_out/trace_forest
