#!/bin/bash
# usage: MESHES="128 256" tools/env_steps.sh "VAR=val" ... : ms/step per mesh and environment
for e in "$@"; do
  for m in ${MESHES:-128 256 512}; do
    env $e python bench.py --mesh $m --no-cpu --no-secondary --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('[$e] mesh $m ms/step', round(d['ms_per_step'],3), 'launches', sum(v['launches'] for v in k.values()), {n: round(v['total_ms'],3) for n,v in k.items() if n.startswith('scan')})"
  done
done
