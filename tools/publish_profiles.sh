#!/bin/bash
# usage: tools/publish_profiles.sh <prefix in gpurun_out, e.g. r2z> [round tag, default r02]  — copies a measurement set into profiles/
set -e
P=$1
R=${2:-r04}
C=$(git rev-parse --short HEAD)
python3 - <<PY
import json
t=json.load(open('gpurun_out/pmc_traffic.json'))
t['commit']='$C'
json.dump(t, open('profiles/${R}_pmc_traffic.json','w'), indent=1)
b=json.loads(open('gpurun_out/${P}_bench.json').read().strip().splitlines()[-1])
b['roofline']['traffic']=t['all_gemm']['hbm_bytes_per_launch']
b['roofline']['traffic_source']='profiles/${R}_pmc_traffic.json (offline rocprofv3 --pmc passes, commit $C)'
open('profiles/${R}_bench_default.json','w').write(json.dumps(b)+"\n")
PY
cp gpurun_out/${P}_shapes.csv profiles/${R}_gemm_shapes.csv
(echo "# ${R} — rocprofv3 --kernel-trace --stats of \`python3 bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline\` (MMSA_BENCH_NOPROF=1: no event / stamp / forward passes)"; echo; echo "MI355X (gfx950), bf16, B=64, S=128, 224x224; 13 steps in the trace (3 warm-up + 10 timed). Commit at capture: $C. Summarised from the rocpd database by tools/prof_summary.py."; echo; python3 tools/prof_summary.py gpurun_out/${P}_prof/run_results.db 13) > profiles/${R}_rocprofv3_kernel_stats.md
echo "published ${R} from ${P} at $C"
