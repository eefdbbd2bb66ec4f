#!/bin/bash
# build a variant of the library into alphazero_amd/lib/dev/<name>.so:  build_variant.sh name [-DFOO=1 ...]
name=$1; shift
cd "$(dirname "$0")/../.." && mkdir -p alphazero_amd/lib/dev
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math "$@" -o alphazero_amd/lib/dev/$name.so alphazero_amd/csrc/az_engine.hip alphazero_amd/csrc/az_tower.hip alphazero_amd/csrc/az_tower_f32.hip alphazero_amd/csrc/az_smallnet.hip alphazero_amd/csrc/az_tower_h2.hip alphazero_amd/csrc/az_tower_h1.hip 2>&1 | grep -E "error|spill" ; ls -la alphazero_amd/lib/dev/$name.so
