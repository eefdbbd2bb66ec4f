#!/bin/bash
# The tree kernel in the regime a trained network produces (peaked priors, deep trees): bench.py --prior-sharpness sweep,
# bf16 tower (where the tree share of a step is largest) and f32 tower.   gpurun -- bash scripts/tree_regimes.sh TAG "1 8 16 32"
set -e -o pipefail
TAG=${1:-r03}
SWEEP=${2:-"1 8 16 32"}
DTYPES=${3:-"bf16 f32"}
# AZ_ENGINE_LIB (alphazero_amd/_build.py) selects another build of the library, e.g. alphazero_amd/lib/dev/r02tree.so (round 2's record layout)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/regimes_$TAG
mkdir -p $OUT
COMMON="--steps 2 --warmup 1 --no-cpu-baseline --game-probe 0 --secondary-steps 0 --no-agreement"
# DTYPES="hash": the tree kernels alone with the synthetic evaluator, SWEEP = its logit scales (--hash-scale)
for S in $SWEEP; do
  for DT in $DTYPES; do
    if [ "$DT" = hash ]; then EV="--evaluator hash --dtype bf16 --hash-scale $S"; else EV="--dtype $DT --prior-sharpness $S"; fi
    timeout -k 10 200 python3 $ROOT/bench.py $COMMON $EV > $OUT/${DT}_s$S.json 2> $OUT/${DT}_s$S.err
    python3 - $OUT/${DT}_s$S.json $DT $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t, k = d["roofline_tree"], d["tree_kernels"]
nn = d.get("nn_eval", {})
print(f"{sys.argv[2]} s={sys.argv[3]}: {d['value']/1e6:.2f} M sims/s, depth {k['mean_depth']:.2f}, children/level {k['mean_children_per_level']:.0f}, "
      f"touched {t['touched_share_of_scanned']:.2f}, k_expand_select {t['ms_per_launch']*1e3:.1f} us, {t['bytes_per_launch']/1e6:.0f} MB, "
      f"{t['achieved']:.0f} GB/s = {t['frac']:.2f}; r02 layout bytes {t['round2_layout']['bytes_per_launch']/1e6:.0f} MB -> {t['round2_layout']['frac_at_this_time']:.2f}; "
      f"tree share of step {1 - nn.get('share_of_step', 0):.3f}, evals/sim {k['evals_per_sim']:.2f}")
PY
  done
done
