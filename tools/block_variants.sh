#!/bin/bash
# usage: tools/block_variants.sh "<flags1>" ... (on the GPU box): rebuild cfx_gather.hip per flag set, time config 5's share
set -eu
BASE="-O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-pass-failed"
for f in "$@"; do
  touch cutfemx_amd/csrc/cfx_gather.hip
  make -C cutfemx_amd/csrc -j8 CXXFLAGS="$BASE $f" > /dev/null 2>&1
  echo "== variant [$f]"
  python tools/time_cfg5.py 2>/dev/null | tail -2
done
