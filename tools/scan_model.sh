#!/bin/bash
# whole-step time against cost-model constants of the GEMM planner (same box): tools/scan_model.sh "a4,b4,a2,b2,c,d" ...
for m in "$@"; do
  MMSA_G2_MODEL=$m MMSA_BENCH_NOPROF=1 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['ms_per_step'])"
done
