#!/bin/bash
set -e
BASE="-O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-pass-failed"
for f in "$@"; do
  touch cutfemx_amd/csrc/cfx_gather.hip
  make -C cutfemx_amd/csrc -j8 CXXFLAGS="$BASE $f" > /dev/null 2>&1
  echo "== variant [$f]"
  python tools/time_p2.py 256 2>/dev/null | grep -v amdgpu | tail -n 3 | cut -c1-260
done
