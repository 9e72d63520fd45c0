set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py > gpurun_out/r4/t_step.log 2>&1; rc=$?
tail -n 30 gpurun_out/r4/t_step.log
exit $rc
