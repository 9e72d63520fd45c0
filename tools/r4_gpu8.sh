#!/bin/bash
set -e
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step.py tests/test_gpu_spaces.py tests/test_gpu_config128.py tests/test_gpu_kernel_paths.py -x -q > $O/t8.log 2>&1 || { tail -40 $O/t8.log; exit 1; }
tail -3 $O/t8.log
python bench.py --n 512 --steps 10 --warmup 2 --no-cpu --no-secondary > $O/b512_f.json 2> $O/b512_f.err || { tail -20 $O/b512_f.err; exit 1; }
CFX_FACET_SORT=1 python bench.py --n 512 --steps 10 --warmup 2 --no-cpu --no-secondary > $O/b512_fs.json 2> $O/b512_fs.err
python bench.py --n 32 --steps 200 --warmup 20 --no-cpu --no-secondary > $O/b32_f.json 2> $O/b32_f.err
python - <<'PY'
import json
for f in ("b512_f","b512_fs","b32_f"):
    d=json.loads(open(f"gpurun_out/r4/{f}.json").read().strip().splitlines()[-1])
    k=d.get("kernels",{})
    print(f, round(d["ms_per_step"],4), d.get("step_mode",{}).get("launches_per_step"), d.get("step_mode",{}).get("read_backs_per_step"),
          {n:round(v["total_ms"],3) for n,v in k.items() if n.startswith("facet_dof")})
PY
