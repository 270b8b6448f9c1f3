#!/bin/bash
# usage: tools/publish_profiles.sh <prefix in gpurun_out, e.g. f2>   — copies a measurement set into profiles/ and refreshes DESIGN.md
set -e
P=$1
cp gpurun_out/pmc_traffic.json profiles/r01_pmc_traffic.json
cp gpurun_out/${P}_shapes.csv profiles/r01_gemm_shapes.csv
python3 - <<PY
import json
b=json.loads(open('gpurun_out/${P}_bench.json').read().strip().splitlines()[-1])
t=json.load(open('profiles/r01_pmc_traffic.json'))['all_gemm']['hbm_bytes_per_launch']
b['roofline']['traffic']=t
open('profiles/r01_bench_default.json','w').write(json.dumps(b)+"\n")
PY
(echo "# Round 1 — rocprofv3 --kernel-trace --stats of \`python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline\` (MMSA_BENCH_NOPROF=1: no event / stamp passes)"; echo; echo "MI355X (gfx950), bf16, B=64, S=128, 224x224; 13 steps in the trace (3 warm-up + 10 timed). Commit at capture: end of round 1. Summarised from the rocpd database by tools/prof_summary.py."; echo; python3 tools/prof_summary.py gpurun_out/${P}_prof/run_results.db 13) > profiles/r01_rocprofv3_kernel_stats.md
python3 tools/fill_design.py profiles/r01_bench_default.json profiles/r01_pmc_traffic.json tools/DESIGN.template.md
