#!/bin/bash
# MFMA-pipe utilisation and effective clock of the tower kernels from PMC counters (on the GPU box: gpurun -- bash scripts/pmc_mfma.sh TAG).
# One --pmc pass (3 SQ + 1 GRBM counters), kernel trace only.  effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time;
# pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8).
set -e -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/mfma_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --preroll 0 --no-cpu-baseline --no-agreement --game-probe 0 ${BENCH_ARGS:-}"
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "k_tower" \
  --output-format csv -d $OUT -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:50]
        a = acc[k][row["Counter_Name"]]
        a[0] += 1; a[1] += float(row["Counter_Value"])
dur = defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:50]
        dur[k][0] += 1; dur[k][1] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
for k, cs in acc.items():
    m = {c: v[1] / max(v[0], 1) for c, v in cs.items()}
    ns = dur[k][1] / max(dur[k][0], 1)
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    clk = gui / 8.0 / ns if ns else 0.0
    util = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * 256 * gui / 8.0) if gui else 0.0
    print(f"{k:50s} launches={dur[k][0]} time={ns/1e3:9.1f} us  clock={clk:5.2f} GHz  mfma_busy={m.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.4g}  "
          f"pipe_util={util:5.3f}  sq_busy={m.get('SQ_BUSY_CYCLES',0):.4g} wave_cycles={m.get('SQ_WAVE_CYCLES',0):.4g}")
PY
