#!/bin/bash
# usage: tools/variant.sh <name> <file.hip> [-DFLAG=..]...: build/libcfx_<name>.so = the engine with <file.hip> recompiled
# with extra defines (timing variants; load it with CFX_LIB=build/libcfx_<name>.so)
set -eu
NAME=$1; FILE=$2; shift 2
cd "$(dirname "$0")/../cutfemx_amd/csrc"
mkdir -p ../../build
OBJS=""
for f in cfx_runtime cfx_mesh cfx_cut cfx_fem cfx_rowasm cfx_gather cfx_dist cfx_f32 cfx_c128 cfx_rtc; do
  if [ "$f.hip" = "$FILE" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -Wno-pass-failed "$@" -c $f.hip -o /tmp/${f}_$NAME.o
    OBJS="$OBJS /tmp/${f}_$NAME.o"
  else
    OBJS="$OBJS $f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libcfx_$NAME.so $OBJS cfx_quadhost.o -ldl
