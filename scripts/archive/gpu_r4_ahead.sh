#!/bin/bash
# round 4 (late): the alpha CG's first iteration(s) ahead of the plane products' gate (key 0 = 64 with key 15 = iterations ahead)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
CFGS="cfg3" bash scripts/gpu_ab.sh 0=0 0=64,15=1 0=64,15=2 0=64,15=3 0=0 0=64,15=1 0=64,15=2 0=64,15=3
CFGS="cfg2 cfg4" bash scripts/gpu_ab.sh 0=0 0=64,15=1 0=0 0=64,15=1
