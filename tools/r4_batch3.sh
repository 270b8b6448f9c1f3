#!/bin/bash
# round-4 batch 3: streaming 1x1 kernel parity, reduce-scatter step, engines; A/B bench of the streaming kernel
set -o pipefail
O=gpurun_out/r4c
mkdir -p $O
python -m pytest tests/test_gemm_stream_gpu.py -x -q > $O/t_stream.log 2>&1; echo "stream tests rc=$?"; tail -3 $O/t_stream.log
python -m pytest tests/test_ddp_gpu.py -q -s > $O/t_ddp.log 2>&1; echo "ddp tests rc=$?"; grep -n "differing\|passed\|failed" $O/t_ddp.log | tail -5
python -m pytest tests/test_engines_gpu.py tests/test_golden_gpu.py -x -q > $O/t_eng.log 2>&1; echo "engine+golden tests rc=$?"; tail -3 $O/t_eng.log
for v in stream1x1 ""; do
  MMSA_DISABLE=$v MMSA_PROF_DUMP=$O/shapes_${v:-on}.csv python3 bench.py --no-cpu-baseline --repeats 3 --exact-steps 0 > $O/ab_stream_${v:-on}.json 2>> $O/ab.err
  python3 -c "import json,sys;d=json.loads(open('$O/ab_stream_${v:-on}.json').read().strip().splitlines()[-1]);print('disable=[$v]',d['protocol']['ms_per_step_by_region'],d['roofline']['kernel_ms_per_step'],d['roofline']['isolated']['kernel_ms_per_step'], d['forward']['ms'])"
done
