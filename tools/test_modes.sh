#!/bin/bash
# the GPU suite in its four assembly modes (full-size cases only in the default mode); logs under gpurun_out/
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/t_default.log 2>&1 || { tail -n 30 gpurun_out/t_default.log; exit 1; }
for mode in "CFX_DETERMINISTIC=1" "CFX_ASSEMBLY=atomic" "CFX_STENCIL=0"; do
  name=$(echo "$mode" | tr -c 'A-Za-z0-9' '_')
  env "$mode" python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > "gpurun_out/t_${name}.log" 2>&1 \
    || { tail -n 30 "gpurun_out/t_${name}.log"; exit 1; }
done
for f in gpurun_out/t_*.log; do echo "$f: $(tail -n 1 "$f")"; done
