#!/bin/bash
# round-4 batch 8: half-wave LayerNorm: kernel / engine / golden tests, A/B bench
set -o pipefail
O=gpurun_out/r4i
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_engines_gpu.py tests/test_golden_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for v in ln_halfwave ""; do
  MMSA_DISABLE=$v python3 bench.py --no-cpu-baseline --repeats 3 --exact-steps 0 > $O/ab_${v:-on}.json 2>> $O/ab.err
  python3 -c "import json,sys;d=json.loads(open('$O/ab_${v:-on}.json').read().strip().splitlines()[-1]);print('disable=[$v]',d['protocol']['ms_per_step_by_region'],d['roofline']['kernel_ms_per_step'],d['roofline']['isolated']['kernel_ms_per_step'], d['forward']['ms'])"
done
