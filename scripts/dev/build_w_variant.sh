#!/bin/bash
# a variant of the library that differs in az_tower_w.hip only: build_w_variant.sh name [-DAZ_W_ABLATE=1 ...] -> alphazero_amd/lib/dev/<name>.so
# (the other objects come from the in-tree build; load it with AZ_ENGINE_LIB=alphazero_amd/lib/dev/<name>.so)
name=$1; shift
cd "$(dirname "$0")/../.." && mkdir -p alphazero_amd/lib/dev
O=alphazero_amd/lib/obj
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math "$@" -c -o alphazero_amd/lib/dev/$name.o alphazero_amd/csrc/az_tower_w.hip 2>&1 | grep -E "error|spill"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o alphazero_amd/lib/dev/$name.so $O/az_engine.hip.o $O/az_tower.hip.o $O/az_tower_f32.hip.o $O/az_smallnet.hip.o $O/az_tower_h2.hip.o $O/az_tower_h1.hip.o alphazero_amd/lib/dev/$name.o && rm -f alphazero_amd/lib/dev/$name.o
ls -la alphazero_amd/lib/dev/$name.so
