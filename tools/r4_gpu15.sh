#!/bin/bash
set -e
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_spaces.py tests/test_gpu_kernel_paths.py tests/test_gpu_facets.py tests/test_gpu_step.py tests/test_gpu_fuzz.py tests/test_gpu_extensions.py tests/test_rectangular_forms.py -x -q > $O/t15.log 2>&1 || { tail -40 $O/t15.log; exit 1; }
tail -2 $O/t15.log
timeout -k 10 600 python tools/time_p2.py > $O/p2_c.txt 2> $O/p2_c.err; tail -5 $O/p2_c.txt | head -3
CFX_PATTERN_SEED=0 timeout -k 10 600 python tools/time_p2.py > $O/p2_c0.txt 2> $O/p2_c0.err; tail -5 $O/p2_c0.txt | head -3
timeout -k 10 600 python tools/time_cfg5.py > $O/cfg5_c.txt 2>&1; tail -2 $O/cfg5_c.txt
