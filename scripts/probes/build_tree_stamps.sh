#!/bin/bash
# Builds a copy of libtakgpu.so whose search kernels carry the s_memtime stamps (-DTG_TREE_STAMPS) into scripts/probes/_bin/
# and runs scripts/probes/tree_stamps.py on it.  Run on the GPU box:  bash scripts/probes/build_tree_stamps.sh [precision] [plies]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/scripts/probes/_bin
mkdir -p $B
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS -DTG_TREE_STAMPS -c $R/tak_amd/csrc/search_kernels.hip -o $B/search_kernels_stamps.o
OBJS=$(ls $R/tak_amd/csrc/_obj/*.o | grep -v search_kernels.o)
/opt/rocm/bin/hipcc $FLAGS -shared -o $B/libtakgpu_stamps.so $OBJS $B/search_kernels_stamps.o -ldl
TAKGPU_LIB=$B/libtakgpu_stamps.so python3 $R/scripts/probes/tree_stamps.py "${1:-f32}" "${2:-12}"
