set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4/t_all.log 2>&1; rc=$?
tail -n 15 gpurun_out/r4/t_all.log
[ $rc -eq 0 ] || exit $rc
CFX_COUNT_SYNC=1 python bench.py --mesh 32 --steps 50 --warmup 5 --no-cpu --no-secondary > gpurun_out/r4/b32s.json 2> gpurun_out/r4/b32s.err
python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary > gpurun_out/r4/b512s.json 2> gpurun_out/r4/b512s.err
tail -c 300 gpurun_out/r4/b32s.err
