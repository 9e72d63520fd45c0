#!/bin/bash
set -e
mkdir -p gpurun_out/r4
O=gpurun_out/r4
K=plan_row_lists,pattern_indptr,vec_plain_offsets,plan_plain_tiles,scan_chained
timeout -k 10 600 python -m pytest tests/test_gpu_step.py tests/test_gpu_parity.py -x -q > $O/t18.log 2>&1 || { tail -40 $O/t18.log; exit 1; }
tail -2 $O/t18.log
bash tools/variant_bench.sh 512 $K -
CFX_FUSED_TILES=1000000 bash tools/variant_bench.sh 512 $K -
CFX_FUSED_TILES=0 bash tools/variant_bench.sh 512 $K -
bash tools/variant_bench.sh 256 $K -
CFX_FUSED_TILES=0 bash tools/variant_bench.sh 256 $K -
bash tools/variant_bench.sh 128 $K -
CFX_FUSED_TILES=0 bash tools/variant_bench.sh 128 $K -
bash tools/variant_bench.sh 32 $K -
CFX_FUSED_TILES=0 bash tools/variant_bench.sh 32 $K -
