#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for V in 8=32,12=1 8=0,12=1 8=0,12=1,11=1 8=0,12=1,11=2; do
  echo "== $V"
  NNGP_DEBUG=$V timeout -k 10 300 python bench.py --config cfg3 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -a "v4 timing" | tail -1
done
