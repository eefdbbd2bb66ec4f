#!/bin/bash
# build + time tower variants on the GPU box:  bash scratch/tower_variants.sh "<defs>" "<defs>" ...
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt $cfg -o alphazero_amd/lib/libaz_engine.so alphazero_amd/csrc/az_engine.hip alphazero_amd/csrc/az_tower.hip 2>/dev/null || { echo "build failed: $cfg"; continue; }
  echo "[$cfg]: $(timeout -k 10 120 python -u scripts/dev/tower_time.py 2>&1 | tail -1)"
done
