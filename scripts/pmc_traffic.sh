#!/bin/bash
# HBM traffic of the tree kernels and the conv tower from PMC counters (run on the GPU box: gpurun -- bash scripts/pmc_traffic.sh TAG).
# Two separate --pmc passes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2), kernel-trace only.
set -e -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --preroll 0 --no-cpu-baseline --no-agreement --game-probe 0 ${BENCH_ARGS:-}"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "k_select_gather|k_expand_backup|k_expand_select|k_tower|k_heads" \
    --output-format csv -d $OUT/$C -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/$C.log 2>&1
  echo "pass $C done"
done
python3 $ROOT/scripts/pmc_summary.py $OUT | tee $OUT/summary.txt
