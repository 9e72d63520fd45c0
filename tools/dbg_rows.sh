#!/bin/bash
for d in 0 2 8 10; do
  echo "== CFX_DEBUG_ROWS=$d"
  CFX_DEBUG_ROWS=$d python bench.py --mesh ${MESH:-256} --no-cpu --no-secondary --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print({n: round(k[n]['avg_us'],1) for n in ('assemble_rows_p1','assemble_rows_cut') if n in k}, d['counts'])
"
done
