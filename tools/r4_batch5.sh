#!/bin/bash
# round-4 batch 5: all GPU tests (no -x); A/B bench of the 3x3 stream
set -o pipefail
O=gpurun_out/r4e
mkdir -p $O
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
for v in stream3x3 ""; do
  MMSA_DISABLE=$v MMSA_PROF_DUMP=$O/shapes_${v:-on}.csv python3 bench.py --no-cpu-baseline --repeats 3 --exact-steps 0 > $O/ab_${v:-on}.json 2>> $O/ab.err
  python3 -c "import json,sys;d=json.loads(open('$O/ab_${v:-on}.json').read().strip().splitlines()[-1]);print('disable=[$v]',d['protocol']['ms_per_step_by_region'],d['roofline']['kernel_ms_per_step'],d['roofline']['isolated']['kernel_ms_per_step'], d['forward']['ms'])"
done
MMSA_PROF_DUMP=$O/shapes_fp32.csv python3 bench.py --precision fp32 --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline > $O/bench_fp32.json 2>> $O/ab.err; cut -c1-300 $O/bench_fp32.json
