#!/bin/bash
# round 4: the 12-pair int8 product alone (3 x 5 planes, cut 4) at the posterior's shape: kernel time under rocprofv3 --stats
cd "$(dirname "$0")/.."
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf $R/gpurun_out/r4w
for P in "3 5 4" "5 5 4"; do
  T=$(echo $P | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4w/$T -o lab -- python3 $R/scripts/i8s_lab.py 1024 32768 32768 $P 3 > $R/gpurun_out/r4w_$T.json 2> $R/gpurun_out/r4w_$T.log
  echo "== planes $P"; cat $R/gpurun_out/r4w_$T.json
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/r4w/$T/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'i8s' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'])
PY
done
find $R/gpurun_out/r4w -name "*.csv" -size +1M -delete
