#!/bin/bash
# timing-ablation builds of assemble_rows_block_p2_kernel (wrong results): build/libcfx_bp2_<bits>.so for CFX_LIB=
set -eu
cd "$(dirname "$0")/../cutfemx_amd/csrc"
mkdir -p ../../build
for bits in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -Wno-pass-failed -DCFX_BP2_ABLATE=$bits -c cfx_gather.hip -o /tmp/cfx_gather_bp2_$bits.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libcfx_bp2_$bits.so cfx_runtime.o cfx_mesh.o cfx_cut.o cfx_fem.o cfx_rowasm.o /tmp/cfx_gather_bp2_$bits.o cfx_dist.o cfx_f32.o cfx_quadhost.o -ldl ) &
done
wait
