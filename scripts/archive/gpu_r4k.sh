#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
CFGS="cfg3" bash scripts/gpu_ab.sh 13=0 13=16 13=32 13=48 13=0 13=32
