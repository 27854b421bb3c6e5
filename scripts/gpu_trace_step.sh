#!/bin/bash
# kernel trace of bench steps; prints the per-queue timeline of the last one.  usage: gpu_trace_step.sh <tag> [bench args...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$TAG -o t -- python3 $R/bench.py --config cfg3 --steps 2 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/trace_$TAG.log 2>&1 || exit 1
F=$(find $R/gpurun_out/trace_$TAG -name "*kernel_trace.csv" | head -1)
python3 $R/scripts/trace_step.py $F 30 > $R/gpurun_out/timeline_$TAG.txt
# keep the merged output small: the raw trace stays on the box
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$F")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "k_row_sqnorm" in r["Kernel_Name"]]
i0 = st[-2] if len(st) >= 2 else st[-1]   # set_train and predict both take row norms: the step starts at the first of the last pair
t0 = int(rows[i0]["Start_Timestamp"])
with open("$R/gpurun_out/timeline_${TAG}.csv", "w") as f:
    f.write("start_us,dur_us,queue,grid,wg,name\n")
    for r in rows[i0:]:
        nm = r["Kernel_Name"].replace("void ", "").replace("nngp::(anonymous namespace)::", "").replace("nngp::", "").split("(")[0][:60]
        f.write("%.1f,%.1f,%s,%s,%s,%s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Queue_Id"], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), nm))
PY
rm -rf $R/gpurun_out/trace_$TAG
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open("$R/gpurun_out/timeline_${TAG}.csv")):
    a = agg[r["name"]]; a[0] += 1; a[1] += float(r["dur_us"])
print("kernel totals of the last step (launches, ms):")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]: print("  %-52s %5d %9.3f" % (k, v[0], v[1] / 1e3))
PY
