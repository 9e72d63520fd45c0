set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py > gpurun_out/r4/t_step.log 2>&1; rc=$?
tail -n 30 gpurun_out/r4/t_step.log
[ $rc -eq 0 ] || exit $rc
python bench.py --mesh 32 --steps 50 --warmup 5 --no-cpu --no-secondary > gpurun_out/r4/b32s.json 2> gpurun_out/r4/b32s.err
python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary > gpurun_out/r4/b512s.json 2> gpurun_out/r4/b512s.err
python3 - <<'PY'
import json
for f in ['b32s','b512s']:
    d=json.loads(open(f'gpurun_out/r4/{f}.json').read().strip().splitlines()[-1])
    k=d['kernels']
    print(f, round(d['ms_per_step'],4), 'launches', sum(v['launches'] for v in k.values()), 'kernel ms', round(sum(v['total_ms'] for v in k.values()),3), d['phases_ms'])
PY
