#!/bin/bash
# round 4, session 6: setup kernels after the staged single-pass builds, tests
set -e
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout -k 10 300 python tools/time_setup.py 512 > $O/setup_new2.txt 2>&1
CFX_STENCIL_STAGED=0 timeout -k 10 300 python tools/time_setup.py 512 > $O/setup_unstaged.txt 2>&1
cat $O/setup_new2.txt $O/setup_unstaged.txt
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/t6.log 2>&1; tail -3 $O/t6.log
