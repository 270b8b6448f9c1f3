#!/bin/bash
# In-situ tile sweep: one short bench.py run per forced tile shape (MMSA_G2_NJ), each dumping the per-launch GEMM table.
# usage (on the GPU box): tools/sweep_tiles.sh <tag>   -> gpurun_out/<tag>_sweep_<cfg>.csv
T=${1:-s}  # optional: MMSA_G2_SPLIT=1 in the environment prices the no-split plans
for cfg in auto 4 3 2 2:2 2:1; do
  if [ "$cfg" = auto ]; then unset MMSA_G2_NJ; else export MMSA_G2_NJ=$cfg; fi
  MMSA_PROF_DUMP=gpurun_out/${T}_sweep_${cfg/:/_}.csv timeout -k 10 200 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_sweep_${cfg/:/_}.json 2> gpurun_out/${T}_sweep_${cfg/:/_}.err || exit 1
  echo "$cfg done: $(cut -c1-120 gpurun_out/${T}_sweep_${cfg/:/_}.json)"
done
