#!/bin/bash
# round-4 batch 17: full GPU suite at the head-unit / 8-wave-attention build, then two default bench lines
set -o pipefail
O=gpurun_out/r4r
mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  python3 bench.py --no-cpu-baseline --repeats 3 > $O/bench_$i.json 2>> $O/bench.err
  python3 -c "import json,sys;d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]);print(d['value'],d['protocol']['ms_per_step_by_region'],d['roofline']['kernel_ms_per_step'], d['forward']['ms'], d.get('exact_mode',{}).get('pairs_per_s'))"
done
