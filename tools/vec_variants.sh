#!/bin/bash
set -e
BASE="-O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-pass-failed"
for f in "$@"; do
  touch cutfemx_amd/csrc/cfx_gather.hip cutfemx_amd/csrc/cfx_fem.hip cutfemx_amd/csrc/cfx_rowasm.hip
  make -C cutfemx_amd/csrc -j8 CXXFLAGS="$BASE $f" > /dev/null 2>&1
  echo "== variant [$f]" | tee -a gpurun_out/vec_variants.log
  python bench.py --mesh ${MESH:-512} --no-cpu --no-secondary --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('ms/step', round(d['ms_per_step'],3), {n: round(k[n]['avg_us'],1) for n in ('pattern_plain_write','plan_plain_masks','pattern_rows','pattern_diag') if n in k})
" | tee -a gpurun_out/vec_variants.log
done
