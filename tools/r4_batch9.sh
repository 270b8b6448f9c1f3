#!/bin/bash
# round-4 batch 9: all GPU tests with the early sum of squares; A/B
set -o pipefail
O=gpurun_out/r4j
mkdir -p $O
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for v in 0 1; do
  MMSA_EARLY_SUMSQ=$v python3 bench.py --no-cpu-baseline --repeats 3 --exact-steps 0 > $O/ab_early$v.json 2>> $O/ab.err
  python3 -c "import json,sys;d=json.loads(open('$O/ab_early$v.json').read().strip().splitlines()[-1]);print('early=$v',d['protocol']['ms_per_step_by_region'],d['roofline']['kernel_ms_per_step'],d['roofline']['isolated']['kernel_ms_per_step'], d['forward']['ms'], d['loss'])"
done
