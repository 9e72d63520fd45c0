#!/bin/bash
# usage: tools/env_variants.sh "VAR=val ..." ... : bench at $MESH with different environments
for e in "$@"; do
  echo "== env [$e]"
  env $e python bench.py --mesh ${MESH:-256} --no-cpu --no-secondary --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('ms/step', round(d['ms_per_step'],3), {n: round(k[n]['avg_us'],1) for n in ('assemble_rows_plain','assemble_rows_p1','assemble_rows_cut','assemble_vec_rows','pattern_rows','pattern_plain') if n in k})
"
done
