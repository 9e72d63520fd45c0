#!/bin/bash
# usage: tools/pmc_any.sh <tag> <kernel-regex> <script.py> "<counters pass 1>" "<counters pass 2>" ...
# one rocprofv3 --pmc pass per counter group over `python3 <script.py>` (run on the GPU box, from the repo root)
set -eu
TAG=$1; RE=$2; SCRIPT=$3; shift 3
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd $R
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --kernel-include-regex "$RE" --pmc $grp --output-format csv -d $R/gpurun_out/pmc_${TAG}_p$i -- python3 $R/$SCRIPT > /dev/null 2> $R/gpurun_out/pmc_${TAG}_p$i.err || echo "pass $i failed"
done
python3 $R/tools/pmc_report.py $R/gpurun_out/pmc_${TAG}_p* > $R/gpurun_out/pmc_${TAG}_summary.txt
