#!/bin/bash
# build + time tower variants on the GPU box
set -e
cd $GRAFT_REPO_ROOT
for cfg in "1 3" "3 2" "1 2"; do
  set -- $cfg
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DAZ_TOWER_STAGE_TAPS=$1 -DAZ_TOWER_WAVES_PER_SIMD=$2 -o alphazero_amd/lib/libaz_engine.so alphazero_amd/csrc/az_engine.hip alphazero_amd/csrc/az_tower.hip 2>/dev/null
  echo "ST=$1 WPS=$2: $(timeout -k 10 120 python -u scratch/tower_time.py 2>&1 | tail -1)"
done
