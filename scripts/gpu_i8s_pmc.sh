#!/bin/bash
# PMC counters of the sliced int8 GEMM (separate passes, kernel-trace only) on the posterior's residual shape 1024 x 32768 x 32768
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/i8pmc_$i -o p -- python3 $R/scripts/i8s_lab.py 1024 32768 32768 5 5 4 1 > $R/gpurun_out/i8pmc_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(list)
dur = []
for f in glob.glob(R + "/gpurun_out/i8pmc_*/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gemm_nt_i8s" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(R + "/gpurun_out/i8pmc_*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gemm_nt_i8s" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("k_gemm_nt_i8s durations under the counters, ms:", [round(d, 2) for d in dur])
for k, v in sorted(agg.items()):
    print(k, "n=%d" % len(v), "mean=%.4g" % (sum(v) / len(v)))
PY
