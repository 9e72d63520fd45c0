#!/bin/bash
# usage: tools/gather_variants.sh "<flags1>" "<flags2>" ...   (run on the GPU box): rebuild cfx_gather.hip per flag set, bench
set -eu
BASE="-O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-pass-failed"
for f in "$@"; do
  touch cutfemx_amd/csrc/cfx_gather.hip
  make -C cutfemx_amd/csrc -j8 CXXFLAGS="$BASE $f" > /dev/null 2>&1
  echo "== variant [$f]"
  python bench.py --mesh ${MESH:-512} --no-cpu --no-secondary --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('ms/step', round(d['ms_per_step'],3), {n: round(k[n]['avg_us'],1) for n in ('assemble_tiles_plain','assemble_rows_plain','assemble_rows_p1','assemble_rows_cut','assemble_vec_plain','vec_tensors_std') if n in k})
"
done
