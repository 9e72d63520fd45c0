#!/bin/bash
# usage: tools/profile_round.sh <tag> <mesh>: rocprofv3 kernel stats + three PMC passes of bench.py (run on the GPU box)
set -eu
TAG=$1; N=$2
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp
export CFX_BENCH_DETAIL=prof_${TAG}_bench_detail.json CFX_BENCH_PRINT_DETAIL=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/bench.py --mesh $N --steps 10 --warmup 2 --no-cpu --no-secondary > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_${TAG}.err
echo stats done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_${TAG}_sq -- python3 $R/bench.py --mesh $N --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $R/gpurun_out/pmc_${TAG}_sq.err
echo sq done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${TAG}_fetch -- python3 $R/bench.py --mesh $N --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $R/gpurun_out/pmc_${TAG}_fetch.err
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_${TAG}_write -- python3 $R/bench.py --mesh $N --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $R/gpurun_out/pmc_${TAG}_write.err
echo write done
python3 $R/tools/pmc_report.py $R/gpurun_out/pmc_${TAG}_sq $R/gpurun_out/pmc_${TAG}_fetch $R/gpurun_out/pmc_${TAG}_write > $R/gpurun_out/pmc_${TAG}_summary.txt
