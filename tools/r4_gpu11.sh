#!/bin/bash
set -e
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_spaces.py tests/test_gpu_kernel_paths.py tests/test_gpu_facets.py tests/test_gpu_fullsize.py -x -q > $O/t11.log 2>&1 || { tail -40 $O/t11.log; exit 1; }
tail -3 $O/t11.log
timeout -k 10 600 python tools/time_p2.py > $O/p2_off.txt 2> $O/p2_off.err; tail -6 $O/p2_off.txt
CFX_OFF_FACET=0 timeout -k 10 600 python tools/time_p2.py > $O/p2_off0.txt 2> $O/p2_off0.err; tail -6 $O/p2_off0.txt | head -3
timeout -k 10 600 python tools/time_cfg5.py > $O/cfg5_off.txt 2>&1; tail -3 $O/cfg5_off.txt
