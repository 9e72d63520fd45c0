set -e
mkdir -p gpurun_out/r4
CFX_COUNT_SYNC=2 python bench.py --mesh 32 --steps 1 --warmup 1 --no-cpu --no-secondary > gpurun_out/r4/sync32.json 2> gpurun_out/r4/sync32.err
python bench.py --mesh 32 --steps 50 --warmup 5 --no-cpu --no-secondary > gpurun_out/r4/b32.json 2> gpurun_out/r4/b32.err
python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary > gpurun_out/r4/b512.json 2> gpurun_out/r4/b512.err
tail -c 600 gpurun_out/r4/b32.json
