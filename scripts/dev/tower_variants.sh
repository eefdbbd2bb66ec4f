#!/bin/bash
# build + time (and optionally test) tower variants on the GPU box:  bash scripts/dev/tower_variants.sh "<defs>" ...
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt $cfg -o alphazero_amd/lib/libaz_engine.so alphazero_amd/csrc/az_engine.hip alphazero_amd/csrc/az_tower.hip 2>/dev/null || { echo "build failed: $cfg"; continue; }
  echo "[$cfg]: $(timeout -k 10 120 python -u scripts/dev/tower_time.py 2>&1 | tail -1)"
  if [ -n "$TOWER_TEST" ]; then timeout -k 10 200 python -u -m pytest tests/test_gpu_tower.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2; fi
done
