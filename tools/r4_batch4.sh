#!/bin/bash
# round-4 batch 4: full GPU tests with the fixed residual epilogue; A/B bench of the streaming kernel
set -o pipefail
O=gpurun_out/r4d
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for v in stream1x1 ""; do
  MMSA_DISABLE=$v MMSA_PROF_DUMP=$O/shapes_${v:-on}.csv python3 bench.py --no-cpu-baseline --repeats 3 --exact-steps 0 > $O/ab_stream_${v:-on}.json 2>> $O/ab.err
  python3 -c "import json,sys;d=json.loads(open('$O/ab_stream_${v:-on}.json').read().strip().splitlines()[-1]);print('disable=[$v]',d['protocol']['ms_per_step_by_region'],d['roofline']['kernel_ms_per_step'],d['roofline']['isolated']['kernel_ms_per_step'], d['forward']['ms'])"
done
