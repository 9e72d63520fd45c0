#!/bin/bash
set -e
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_step.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -x -q > $O/t17.log 2>&1 || { tail -40 $O/t17.log; exit 1; }
tail -2 $O/t17.log
timeout -k 10 200 python tools/launch_trace.py 32 2> $O/launches32b.txt >/dev/null
awk '/==== step 2/{p=1} p' $O/launches32b.txt | wc -l
bash tools/variant_bench.sh 32 plan_row_lists,pattern_indptr,vec_plain_offsets,plan_plain_tiles - 
CFX_FUSED_SCANS=0 bash tools/variant_bench.sh 32 plan_row_lists,pattern_indptr,vec_plain_offsets,plan_plain_tiles -
bash tools/variant_bench.sh 512 plan_row_lists,pattern_indptr,vec_plain_offsets,plan_plain_tiles - 
CFX_FUSED_SCANS=0 bash tools/variant_bench.sh 512 plan_row_lists,pattern_indptr,vec_plain_offsets,plan_plain_tiles -
CFX_INDPTR_CHAINED_TILES=100000 bash tools/variant_bench.sh 512 plan_row_lists,pattern_indptr,vec_plain_offsets,plan_plain_tiles -
