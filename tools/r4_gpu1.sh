set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_user_integrands.py tests/test_gpu_step.py tests/test_gpu_spaces.py -x -q -m gpu > gpurun_out/r4/t_step.log 2>&1; rc=$?
tail -n 40 gpurun_out/r4/t_step.log
exit $rc
