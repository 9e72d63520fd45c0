#!/bin/bash
set -e
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step.py tests/test_gpu_spaces.py tests/test_gpu_facets.py tests/test_gpu_kernel_paths.py tests/test_rectangular_forms.py tests/test_gpu_extensions.py -x -q > $O/t13.log 2>&1 || { tail -40 $O/t13.log; exit 1; }
tail -2 $O/t13.log
CFX_DETERMINISTIC=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_spaces.py -x -q > $O/t13d.log 2>&1 || { tail -40 $O/t13d.log; exit 1; }
tail -2 $O/t13d.log
timeout -k 10 600 python tools/time_p2.py > $O/p2_u.txt 2> $O/p2_u.err; tail -2 $O/p2_u.txt
python bench.py --n 512 --steps 10 --warmup 2 --no-cpu --no-secondary > $O/b512_u.json 2> $O/b512_u.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/b512_u.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("b512_u", round(d["ms_per_step"],4), {n:round(v["total_ms"],3) for n,v in k.items() if n.startswith("facet_dof")})
PY
