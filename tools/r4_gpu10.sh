#!/bin/bash
set -e
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_rectangular_forms.py tests/test_complex_assembly.py -x -q > $O/t10.log 2>&1 || { tail -60 $O/t10.log; exit 1; }
tail -3 $O/t10.log
CFX_ASSEMBLY=atomic timeout -k 10 600 python -m pytest tests/test_rectangular_forms.py -x -q > $O/t10a.log 2>&1 || { tail -40 $O/t10a.log; exit 1; }
tail -2 $O/t10a.log
CFX_DETERMINISTIC=1 timeout -k 10 600 python -m pytest tests/test_rectangular_forms.py -x -q > $O/t10d.log 2>&1 || { tail -40 $O/t10d.log; exit 1; }
tail -2 $O/t10d.log
