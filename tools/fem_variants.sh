#!/bin/bash
# usage: tools/fem_variants.sh <script.py> "<flags1>" ...  (GPU box): rebuild cfx_fem.hip per flag set, run the script
set -eu
SCRIPT=$1; shift
BASE="-O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-pass-failed"
for f in "$@"; do
  touch cutfemx_amd/csrc/cfx_fem.hip
  make -C cutfemx_amd/csrc -j8 CXXFLAGS="$BASE $f" > /dev/null 2>&1
  echo "== variant [$f]"
  python $SCRIPT 2>/dev/null | tail -2
done
