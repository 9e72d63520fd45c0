#!/bin/bash
# usage: tools/pmc2.sh <tag> <mesh> <kernel-regex> "<counters pass 1>" "<counters pass 2>" ...
# one rocprofv3 --pmc pass per counter group (counters only, kernel trace for names), bench.py as workload
set -eu
TAG=$1; N=$2; RE=$3; shift 3
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --kernel-include-regex "$RE" --pmc $grp --output-format csv -d $R/gpurun_out/pmc_${TAG}_p$i -- python3 $R/bench.py --mesh $N --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $R/gpurun_out/pmc_${TAG}_p$i.err || echo "pass $i failed"
done
python3 $R/tools/pmc_report.py $R/gpurun_out/pmc_${TAG}_p* > $R/gpurun_out/pmc_${TAG}_summary.txt
