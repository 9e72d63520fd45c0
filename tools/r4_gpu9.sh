#!/bin/bash
set -e
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step.py tests/test_gpu_spaces.py tests/test_gpu_config128.py tests/test_gpu_kernel_paths.py tests/test_gpu_facets.py tests/test_gpu_edge_cases.py -x -q > $O/t9.log 2>&1 || { tail -40 $O/t9.log; exit 1; }
tail -3 $O/t9.log
CFX_STEP_SPECULATE=0 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step.py -x -q > $O/t9b.log 2>&1 || { tail -40 $O/t9b.log; exit 1; }
tail -2 $O/t9b.log
python bench.py --n 512 --steps 10 --warmup 2 --no-cpu --no-secondary > $O/b512_g.json 2> $O/b512_g.err || { tail -20 $O/b512_g.err; exit 1; }
python bench.py --n 32 --steps 200 --warmup 20 --no-cpu --no-secondary > $O/b32_g.json 2> $O/b32_g.err
python - <<'PY'
import json
for f in ("b512_g","b32_g"):
    d=json.loads(open(f"gpurun_out/r4/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"],4), d.get("step_mode",{}).get("launches_per_step"), d.get("step_mode",{}).get("read_backs_per_step"),
          (d.get("projected_scaling") or {}).get("by_world",{}).get("8"))
PY
