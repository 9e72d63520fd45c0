set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py > gpurun_out/r4/t_step.log 2>&1; rc=$?
tail -n 8 gpurun_out/r4/t_step.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/r4/full.json 2> gpurun_out/r4/full.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/full.json').read().strip().splitlines()[-1])
k=d['kernels']
print(d['ms_per_step'], d['value'], 'launches', sum(v['launches'] for v in k.values()), 'kernel ms', round(sum(v['total_ms'] for v in k.values()),3))
print(json.dumps(d.get('projected_scaling',{}).get('by_world')))
print(d.get('config_128'))
PY
python bench.py --mesh 32 --steps 50 --warmup 5 --no-cpu --no-secondary > gpurun_out/r4/b32s.json 2>/dev/null
python3 -c "
import json
d=json.loads(open('gpurun_out/r4/b32s.json').read().strip().splitlines()[-1]); k=d['kernels']
print('32^3', d['ms_per_step'], 'launches', sum(v['launches'] for v in k.values()))"
