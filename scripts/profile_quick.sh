#!/bin/bash
# Kernel-time profile + PMC HBM traffic of config 5's QuickModel tower (bench.py --workload gomoku15_quick at a reduced number of
# simulations per move: the tower launch itself is the full 16,384-slot batch):
#   gpurun -- bash scripts/profile_quick.sh TAG  -> gpurun_out/prof_TAG/, gpurun_out/pmc_TAG/
set -e -o pipefail
TAG=${1:-quick}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
ARGS="--workload gomoku15_quick --sims 24 --steps 1 --warmup 0 --preroll 8 --no-cpu-baseline --game-probe 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -o bench -- \
  python3 $ROOT/bench.py $ARGS > $ROOT/gpurun_out/prof_$TAG.log 2>&1
echo "kernel trace done"
cd $ROOT && BENCH_ARGS="--workload gomoku15_quick --sims 12" bash scripts/pmc_traffic.sh $TAG
