#!/bin/bash
set -e
mkdir -p gpurun_out/r4
O=gpurun_out/r4
K=plan_row_lists,pattern_indptr,vec_plain_offsets,plan_plain_tiles,scan_chained
timeout -k 10 600 python -m pytest tests/test_gpu_step.py tests/test_gpu_parity.py -x -q > $O/t19.log 2>&1 || { tail -40 $O/t19.log; exit 1; }
tail -2 $O/t19.log
for n in 512 256 128 64 32; do
bash tools/variant_bench.sh $n $K -
CFX_FUSED_TILES=1000000 bash tools/variant_bench.sh $n $K -
CFX_FUSED_TILES=0 bash tools/variant_bench.sh $n $K -
done
