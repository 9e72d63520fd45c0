#!/bin/bash
set -e
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout -k 10 600 python tools/time_p2.py > $O/p2_s.txt 2> $O/p2_s.err; tail -5 $O/p2_s.txt
CFX_OFF_FACET=0 timeout -k 10 600 python tools/time_p2.py > $O/p2_s0.txt 2> $O/p2_s0.err; tail -5 $O/p2_s0.txt
timeout -k 10 900 python -m pytest tests/test_gpu_spaces.py tests/test_gpu_kernel_paths.py tests/test_gpu_facets.py -x -q > $O/t12.log 2>&1 || { tail -40 $O/t12.log; exit 1; }
tail -2 $O/t12.log
