#!/bin/bash
# round 4, session 5: setup-kernel variants, three row classes of the degree-2 interface kernel
set -e
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout -k 10 300 python tools/time_setup.py 512 > $O/setup_new.txt 2>&1
CFX_ADJ_LDS=0 CFX_C2C=00 timeout -k 10 300 python tools/time_setup.py 512 > $O/setup_old.txt 2>&1
CFX_C2C=01 timeout -k 10 300 python tools/time_setup.py 512 > $O/setup_gather_xcd.txt 2>&1
CFX_C2C=10 timeout -k 10 300 python tools/time_setup.py 512 > $O/setup_lists_noxcd.txt 2>&1
cat $O/setup_new.txt $O/setup_old.txt $O/setup_gather_xcd.txt $O/setup_lists_noxcd.txt
CFX_PLAN_DEBUG=1 timeout -k 10 600 python tools/time_p2.py > $O/p2_classes.txt 2> $O/p2_classes.err
grep -h "hashed row classes" $O/p2_classes.err | tail -2
tail -8 $O/p2_classes.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_spaces.py tests/test_gpu_config128.py -x -q > $O/t5.log 2>&1; tail -3 $O/t5.log
