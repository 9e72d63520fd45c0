#!/bin/bash
# usage: tools/variant_bench.sh <mesh> <kernel,kernel,...> <lib|-> ...: ms per step and the named kernels' ms (bench.py's
# profiled steps) for each library variant ("-" = the in-tree build); run on the GPU box
set -u
N=$1; K=$2; shift 2
for L in "$@"; do
  if [ "$L" = "-" ]; then unset CFX_LIB; else export CFX_LIB=$L; fi
  timeout -k 10 240 python3 bench.py --mesh $N --steps 4 --warmup 1 --no-cpu --no-secondary 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
ks = '$K'.split(',')
print('$L'.ljust(28), 'step %.2f' % d['ms_per_step'], ' '.join('%s %.3f' % (k, d['kernels'].get(k, {}).get('total_ms', float('nan'))) for k in ks), flush=True)
" || echo "$L failed"
done
