set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4/t_all.log 2>&1; rc=$?
tail -n 12 gpurun_out/r4/t_all.log
exit $rc
