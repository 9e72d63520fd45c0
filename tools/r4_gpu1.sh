set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_parity.py tests/test_gpu_spaces.py tests/test_gpu_edge_cases.py tests/test_gpu_facets.py tests/test_gpu_vec_blocks.py tests/test_complex_assembly.py tests/test_gpu_f32.py tests/test_rectangular_forms.py -x -q -m gpu > gpurun_out/r4/t_step.log 2>&1; rc=$?
tail -n 40 gpurun_out/r4/t_step.log
exit $rc
