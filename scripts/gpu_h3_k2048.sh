#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for M in 30720 16384; do
M=$M timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/h3k2048_$M -o t -- python3 $R/scripts/h3_k2048.py > $R/gpurun_out/h3k2048.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
rows=[r for r in csv.DictReader(open(glob.glob("$R/gpurun_out/h3k2048_$M/**/*kernel_trace.csv", recursive=True)[0])) if "k_gemm_nt_h3" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
k1=sorted(d[0::2])[len(d)//4]; k2=sorted(d[1::2])[len(d)//4]
print("m=$M  K=1024: %.1f us   K=2048: %.1f us   2 x K=1024 / K=2048 = %.3f" % (k1, k2, 2*k1/k2))
PY
done
