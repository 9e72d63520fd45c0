set -o pipefail
export TMPDIR=/tmp
R=$(pwd)
mkdir -p $R/gpurun_out/r4
python3 bench.py --mesh 512 --steps 10 --warmup 2 --no-cpu --no-secondary > gpurun_out/r4/s1.json 2>/dev/null
CFX_BENCH_STREAM=0 python3 bench.py --mesh 512 --steps 10 --warmup 2 --no-cpu --no-secondary > gpurun_out/r4/s0.json 2>/dev/null
python3 bench.py --mesh 32 --steps 50 --warmup 5 --no-cpu --no-secondary > gpurun_out/r4/s1_32.json 2>/dev/null
CFX_BENCH_STREAM=0 python3 bench.py --mesh 32 --steps 50 --warmup 5 --no-cpu --no-secondary > gpurun_out/r4/s0_32.json 2>/dev/null
python3 - <<'PY'
import json
for f in ['s1','s0','s1_32','s0_32']:
    d=json.loads(open(f'gpurun_out/r4/{f}.json').read().strip().splitlines()[-1])
    k=d['kernels']
    print(f, round(d['ms_per_step'],4), 'kernel ms', round(sum(v['total_ms'] for v in k.values()),3))
PY
cd /tmp
rm -rf $R/gpurun_out/r4/trace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4/trace -- python3 $R/bench.py --mesh 512 --steps 4 --warmup 2 --no-cpu --no-secondary > $R/gpurun_out/r4/trace_bench.json 2> $R/gpurun_out/r4/trace.err
python3 $R/tools/trace_gaps.py $R/gpurun_out/r4/trace 7
