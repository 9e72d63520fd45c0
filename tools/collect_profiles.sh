#!/bin/bash
# usage: tools/collect_profiles.sh <tag>: after `gpurun -- tools/profile_round.sh <tag> 512 && tools/profile_cfg45.sh <tag>`
# copy the summaries that DESIGN.md / bench.py quote from gpurun_out/ (scratch) into profiles/ (tracked)
set -eu
TAG=$1
cd "$(dirname "$0")/.."
strip() { sed -e 's#/tmp/code/[^ ]*/repo/##g' -e 's#/root/repo/##g' "$1" > "$2"; }
strip gpurun_out/pmc_${TAG}_summary.txt profiles/${TAG}_n512_pmc_summary.txt
cp "$(find gpurun_out/prof_${TAG} -name '*kernel_stats.csv' -printf '%T@ %p\n' | sort -n | tail -1 | cut -d' ' -f2-)" profiles/${TAG}_n512_kernel_stats.csv   # (the newest: gpurun_out keeps earlier runs)
cp gpurun_out/prof_${TAG}_bench_detail.json profiles/${TAG}_n512_bench_under_rocprof.json
python3 tools/pmc_traffic.py profiles/${TAG}_n512_pmc_summary.txt 512 ${TAG}
for c in cfg4 cfg5; do
  strip gpurun_out/pmc_${TAG}_${c}_summary.txt profiles/${TAG}_${c}_pmc_summary.txt
  cp gpurun_out/prof_${TAG}_${c}_kernel_stats.csv profiles/${TAG}_${c}_kernel_stats.csv
  cp gpurun_out/prof_${TAG}_${c}_phases.txt profiles/${TAG}_${c}_phases.txt
done
python3 tools/pmc_phase_traffic.py profiles/${TAG}_cfg4_pmc_summary.txt 3 profiles/${TAG}_cfg4_traffic.json cfg4 "^assemble_rows" "cut_tensors_p2" "assemble_cells_kernel" "cut_moments" "facet_jumps" "zero_inactive"
python3 tools/pmc_phase_traffic.py profiles/${TAG}_cfg4_pmc_summary.txt 4 profiles/${TAG}_cfg4_sparsity_traffic.json cfg4_sparsity "^pattern_" "indptr_"
python3 tools/pmc_phase_traffic.py profiles/${TAG}_cfg5_pmc_summary.txt 4 profiles/${TAG}_cfg5_traffic.json cfg5 "^assemble_rows_block" "elasticity_tensors" "facet_jumps" "zero_inactive"
python3 tools/pmc_phase_traffic.py profiles/${TAG}_cfg5_pmc_summary.txt 4 profiles/${TAG}_cfg5_sparsity_traffic.json cfg5_sparsity "^pattern_" "indptr_"
