#!/bin/bash
set -e
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_multi_level_set.py tests/test_gpu_f32.py tests/test_gpu_edge_cases.py tests/test_gpu_step.py tests/test_gpu_config128.py tests/test_gpu_facets.py tests/test_gpu_dist.py -x -q > $O/t16.log 2>&1 || { tail -40 $O/t16.log; exit 1; }
tail -2 $O/t16.log
python bench.py --n 512 --steps 10 --warmup 2 --no-cpu --no-secondary > $O/b512_cull.json 2> $O/b512_cull.err || { tail -20 $O/b512_cull.err; exit 1; }
CFX_CLASSIFY_CULL=0 python bench.py --n 512 --steps 10 --warmup 2 --no-cpu --no-secondary > $O/b512_cull0.json 2> $O/b512_cull0.err
python tools/show_bench.py $O/b512_cull.json $O/b512_cull0.json
python - <<'PY'
import json
for f in ("b512_cull","b512_cull0"):
    d=json.loads(open(f"gpurun_out/r4/{f}.json").read().strip().splitlines()[-1])
    print(f, d["phases_ms"]["cut"], {k:round(v["total_ms"],3) for k,v in d["kernels"].items() if k in ("classify","sign_codes","locate_entities")}, d["setup"]["split_ms"])
PY
