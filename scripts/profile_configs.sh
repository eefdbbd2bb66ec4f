#!/bin/bash
# Kernel-time profiles of the other BASELINE configurations (gomoku 9x9, dots-and-boxes, tic-tac-toe), f32 network arithmetic:
#   gpurun -- bash scripts/profile_configs.sh TAG   -> gpurun_out/prof_TAG_<workload>/bench_kernel_stats.csv + the bench line
set -e -o pipefail
TAG=${1:-cfg}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for W in gomoku9 dab tictactoe; do
  G=1; [ $W = gomoku9 ] && G=0
  mkdir -p $ROOT/gpurun_out/prof_${TAG}_$W
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_$W -o bench -- \
    python3 $ROOT/bench.py --workload $W --graph $G --steps 4 --warmup 1 --secondary-steps 0 --no-cpu-baseline --no-agreement --game-probe 0 \
    > $ROOT/gpurun_out/prof_${TAG}_$W.log 2>&1
  echo "$W done"
done
