#!/bin/bash
# round-4 batch 10: all GPU tests; A/B of the deferred finalizes and of the 8-channel pools
set -o pipefail
O=gpurun_out/r4k
mkdir -p $O
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for v in defer_finalize pool8 ""; do
  MMSA_DISABLE=$v python3 bench.py --no-cpu-baseline --repeats 3 --exact-steps 0 > $O/ab_${v:-on}.json 2>> $O/ab.err
  python3 -c "import json,sys;d=json.loads(open('$O/ab_${v:-on}.json').read().strip().splitlines()[-1]);print('disable=[$v]',d['protocol']['ms_per_step_by_region'],d['roofline']['kernel_ms_per_step'],d['roofline']['isolated']['kernel_ms_per_step'], d['forward']['ms'], d['loss'])"
done
