#!/bin/bash
# PMC counters of the float64 MFMA GEMM on the posterior's residual shape (separate passes, kernel-trace only)
export TMPDIR=/tmp F64_VARIANTS=0
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/f64pmc_$i -o p -- python3 $R/scripts/f64_lab.py > $R/gpurun_out/f64pmc_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(list)
for f in glob.glob(R + "/gpurun_out/f64pmc_*/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gemm_nt_f64" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, "n=%d" % len(v), "mean=%.5g" % (sum(v) / len(v)))
PY
