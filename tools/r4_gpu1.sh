set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py > gpurun_out/r4/t_step.log 2>&1; rc=$?
tail -n 30 gpurun_out/r4/t_step.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/r4/full.json 2> gpurun_out/r4/full.err; rc=$?
tail -n 5 gpurun_out/r4/full.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/full.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d.get('step_mode'))
print(json.dumps(d.get('moving_domain'))[:1500])
print(json.dumps(d.get('projected_scaling'))[:1500])
for k in ('config_128','config_p2_gyroid_256','config_elasticity_share'):
    v=d.get(k) or {}
    print(k, v.get('ms_per_step'), v.get('error'), v.get('phases_ms'))
PY
