#!/bin/bash
# usage: tools/profile_cfg45.sh <tag>: rocprofv3 kernel stats + three PMC passes (SQ, FETCH, WRITE/TCC) of the
# configs[3] (tools/time_p2.py) and configs[4]-share (tools/time_cfg5.py) timing programs.  Run on the GPU box from the
# repo root; the program comes directly after `--` (no env / bash hop under the profiler).
set -eu
TAG=$1
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd $R
for cfg in cfg4:tools/time_p2.py cfg5:tools/time_cfg5.py; do
  name=${cfg%%:*}; script=${cfg##*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_${name} -- python3 $R/$script > $R/gpurun_out/prof_${TAG}_${name}_phases.txt 2> $R/gpurun_out/prof_${TAG}_${name}.err
  echo "$name stats done"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_${TAG}_${name}_sq -- python3 $R/$script > /dev/null 2> $R/gpurun_out/pmc_${TAG}_${name}_sq.err
  echo "$name sq done"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${TAG}_${name}_fetch -- python3 $R/$script > /dev/null 2> $R/gpurun_out/pmc_${TAG}_${name}_fetch.err
  echo "$name fetch done"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_${TAG}_${name}_write -- python3 $R/$script > /dev/null 2> $R/gpurun_out/pmc_${TAG}_${name}_write.err
  echo "$name write done"
  python3 $R/tools/pmc_report.py $R/gpurun_out/pmc_${TAG}_${name}_sq $R/gpurun_out/pmc_${TAG}_${name}_fetch $R/gpurun_out/pmc_${TAG}_${name}_write > $R/gpurun_out/pmc_${TAG}_${name}_summary.txt
  cp $(find $R/gpurun_out/prof_${TAG}_${name} -name '*kernel_stats.csv' | head -1) $R/gpurun_out/prof_${TAG}_${name}_kernel_stats.csv
done
