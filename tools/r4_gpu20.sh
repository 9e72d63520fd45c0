#!/bin/bash
set -e
mkdir -p gpurun_out/r4
O=gpurun_out/r4
K=plan_row_lists,pattern_indptr,vec_plain_offsets,plan_plain_tiles,scan_chained,scan_reduce,scan_write,scan_top
timeout -k 10 600 python -m pytest tests/test_gpu_step.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -x -q > $O/t20.log 2>&1 || { tail -40 $O/t20.log; exit 1; }
tail -2 $O/t20.log
timeout -k 10 200 python tools/launch_trace.py 32 2> $O/launches32c.txt >/dev/null
for n in 512 256 64 32; do
bash tools/variant_bench.sh $n $K -
CFX_FUSED_TILES=0 bash tools/variant_bench.sh $n $K -
CFX_SCAN_CHAINED_TILES=1024 bash tools/variant_bench.sh $n $K -
done
