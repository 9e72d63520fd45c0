#!/bin/bash
# usage: tools/pmc.sh <tag> <n> : three rocprofv3 counter passes (SQ, FETCH, WRITE/TCC) of bench.py
set -eu
TAG=$1; N=$2
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_${TAG}_sq -- python3 $R/bench.py --mesh $N --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $R/gpurun_out/pmc_${TAG}_sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${TAG}_fetch -- python3 $R/bench.py --mesh $N --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $R/gpurun_out/pmc_${TAG}_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_${TAG}_write -- python3 $R/bench.py --mesh $N --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $R/gpurun_out/pmc_${TAG}_write.err
ls -R $R/gpurun_out/pmc_${TAG}_sq | head
