set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_gpu_step.py -x -q -m gpu > gpurun_out/r4/t_step.log 2>&1 || { tail -n 30 gpurun_out/r4/t_step.log; exit 1; }
for i in 1 2; do
CFX_BENCH_STEP=0 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary > gpurun_out/r4/c512_exact_$i.json 2> /dev/null
CFX_BENCH_STEP=1 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary > gpurun_out/r4/c512_step_$i.json 2> /dev/null
done
python3 - <<'PY'
import json
for f in ['exact_1','step_1','exact_2','step_2']:
    d=json.loads(open(f'gpurun_out/r4/c512_{f}.json').read().strip().splitlines()[-1])
    k=d['kernels']
    print(f, round(d['ms_per_step'],3), 'kernels', round(sum(v['total_ms'] for v in k.values()),3), {n: round(k[n]['total_ms'],3) for n in ('vec_tensors_std','assemble_tiles_plain','assemble_rows_cut','pattern_plain_write','classify','locate_entities')})
PY
