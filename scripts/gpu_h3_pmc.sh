#!/bin/bash
# PMC counters of the split-float16 GEMM (separate passes, kernel-trace only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/h3pmc_$i -o p -- python3 $R/scripts/h3_one.py > $R/gpurun_out/h3pmc_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(list)
for f in glob.glob(R + "/gpurun_out/h3pmc_*/**/*counter_collection*.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_gemm_nt_h3" in r["Kernel_Name"]]
    by_counter = collections.defaultdict(list)
    for r in rows:
        by_counter[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in by_counter.items():
        v.sort()
        half = len(v) // 2  # scripts/h3_one.py: first the K = 1024 launches, then the K = 4096 ones
        agg[("K1024", c)] += [x for _, x in v[:half]]
        agg[("K4096", c)] += [x for _, x in v[half:]]
for k, v in sorted(agg.items()):
    print(k[0], k[1], "n=%d" % len(v), "last=%.4g" % v[-1], "mean=%.4g" % (sum(v) / len(v)))
PY
