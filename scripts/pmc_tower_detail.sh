#!/bin/bash
# Where a tower kernel's wave time goes, from PMC counters (on the GPU box: gpurun -- bash scripts/pmc_tower_detail.sh TAG).
# Two separate --pmc passes of <= 8 SQ (+ 1 GRBM) counters each, kernel trace only:
#   pass A: MFMA pipe busy, wave cycles split into parked (s_waitcnt / barrier) / issue-stalled / issuing, LDS issue stalls
#   pass B: LDS array cycles and bank-conflict cycles, LDS / VALU / MFMA instruction counts, VALU+MFMA co-execution
# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x4 = cycles), SQ_VALU_MFMA_BUSY_CYCLES counts cycles
# (MI355X_MICROARCH.md).  BENCH_ARGS selects the run, e.g. BENCH_ARGS="--dtype bf16 --secondary-steps 0".
set -e -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${DETAIL_PREFIX:-tower}_detail_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --preroll 0 --no-cpu-baseline --no-agreement --game-probe 0 --secondary-steps 0 ${BENCH_ARGS:-}"
PASS_A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
PASS_B="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
for P in A B; do
  V=PASS_$P
  timeout -s KILL 300 rocprofv3 --pmc ${!V} --kernel-trace --kernel-include-regex "${KERNEL_REGEX:-k_tower}" \
    --output-format csv -d $OUT/$P -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/run_$P.log 2>&1
  echo "pass $P done"
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
dur = defaultdict(lambda: [0, 0.0])
for p in "AB":
    for f in glob.glob(os.path.join(root, p, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][:60]
            a = acc[k][row["Counter_Name"]]
            a[0] += 1; a[1] += float(row["Counter_Value"])
    for f in glob.glob(os.path.join(root, p, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][:60]
            dur[k][0] += 1; dur[k][1] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
for k, cs in acc.items():
    m = {c: v[1] / max(v[0], 1) for c, v in cs.items()}
    ns = dur[k][1] / max(dur[k][0], 1)
    gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # per-XCD cycles of the launch
    simd_cycles = 4 * 256 * gui
    wc = 4.0 * m.get("SQ_WAVE_CYCLES", 0.0)
    print(f"== {k}: {dur[k][0]} launches (both passes), {ns/1e3:.1f} us per launch, clock {gui/ns if ns else 0:.2f} GHz")
    print(f"   MFMA pipe busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/simd_cycles:.3f} of SIMD cycles; MFMA instructions {m.get('SQ_INSTS_MFMA',0):.4g}; "
          f"VALU+MFMA co-execution {m.get('SQ_VALU_MFMA_COEXEC_CYCLES',0)/simd_cycles:.3f}")
    if wc:
        print(f"   wave cycles {wc:.4g} = {wc/simd_cycles:.2f} resident waves per SIMD; of them parked (s_waitcnt / barrier) "
              f"{4*m.get('SQ_WAIT_ANY',0)/wc:.3f}, issue-stalled {4*m.get('SQ_WAIT_INST_ANY',0)/wc:.3f} (LDS issue {4*m.get('SQ_WAIT_INST_LDS',0)/wc:.3f}), "
              f"issuing {4*m.get('SQ_ACTIVE_INST_ANY',0)/wc:.3f}")
    cu_cycles = 256 * gui
    print(f"   LDS array active {m.get('SQ_LDS_IDX_ACTIVE',0)/cu_cycles:.3f} of CU cycles, bank-conflict cycles {m.get('SQ_LDS_BANK_CONFLICT',0)/cu_cycles:.3f}; "
          f"LDS instructions {m.get('SQ_INSTS_LDS',0):.4g}, VALU instructions {m.get('SQ_INSTS_VALU',0):.4g}, "
          f"VALU issue (quad-cycles x4 / SIMD cycles) {4*m.get('SQ_ACTIVE_INST_VALU',0)/simd_cycles:.3f}")
PY
