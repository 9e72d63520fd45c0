#!/bin/bash
set -e
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_gpu_config128.py -x -q > $O/t22.log 2>&1 || { tail -40 $O/t22.log; exit 1; }
tail -2 $O/t22.log
bash tools/variant_bench.sh 512 pattern_indptr,plan_row_lists -
bash tools/variant_bench.sh 512 pattern_indptr,plan_row_lists -
