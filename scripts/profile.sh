#!/bin/bash
# Kernel-time profile (the default bench command, minus the CPU baseline's worker processes) + PMC HBM traffic, on the GPU box:
#   gpurun -- bash scripts/profile.sh TAG      -> gpurun_out/prof_TAG/ (kernel stats), gpurun_out/pmc_TAG/ (traffic)
set -e -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -o bench -- \
  python3 $ROOT/bench.py --no-cpu-baseline --no-agreement --game-probe 0 > $ROOT/gpurun_out/prof_$TAG.log 2>&1
echo "kernel trace done"
cd $ROOT && bash scripts/pmc_traffic.sh $TAG
