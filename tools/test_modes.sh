#!/bin/bash
# the GPU suite in its assembly modes and with the round-2 specialisations switched off one by one (full-size cases and
# the kernel-path checks only in the default mode); logs under gpurun_out/
set -o pipefail
mkdir -p gpurun_out
# usage: tools/test_modes.sh [first last]: only the modes number first..last of the list (0 = the default run); a whole
# pass is longer than one gpurun call may last
FIRST=${1:-0}; LAST=${2:-999}
if [ "$FIRST" -le 0 ]; then
python -m pytest tests -m gpu -x -q > gpurun_out/t_default.log 2>&1 || { tail -n 30 gpurun_out/t_default.log; exit 1; }
fi
K=0
for mode in "CFX_DETERMINISTIC=1" "CFX_ASSEMBLY=atomic" "CFX_STENCIL=0" "CFX_STENCIL_LISTS=0" "CFX_P2_PLAIN=0" "CFX_P2_CLOSED=0" \
            "CFX_P2_MOMENTS=0" "CFX_FACET_FOLD_STAGE1=0" "CFX_BLOCK_PLAIN=0" "CFX_TILES=0" "CFX_LAZY_ZERO=0" "CFX_P2_CUT_TENSORS=0" \
            "CFX_P2_INTERFACE=0" "CFX_VEC_BLOCKS=0" "CFX_VEC_BLOCKS=2" "CFX_STEP_SPECULATE=0" "CFX_FACET_SORT=1" "CFX_RECT_GATHER=0" "CFX_CLASSIFY_CULL=0" "CFX_FUSED_TILES=0" "CFX_FUSED_TILES=1000000" "CFX_BULK_ROWS=0"; do
  K=$((K+1))
  if [ "$K" -lt "$FIRST" ] || [ "$K" -gt "$LAST" ]; then continue; fi
  name=$(echo "$mode" | tr -c 'A-Za-z0-9' '_')
  env "$mode" python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_kernel_paths.py \
    --deselect tests/test_gpu_config128.py::test_cfg128_takes_the_specialised_kernels > "gpurun_out/t_${name}.log" 2>&1 \
    || { tail -n 30 "gpurun_out/t_${name}.log"; exit 1; }
done
for f in gpurun_out/t_*.log; do echo "$f: $(tail -n 1 "$f")"; done
