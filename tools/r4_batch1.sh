#!/bin/bash
# round-4 batch 1: GPU tests, default bench, A/B of the grouped weight-gradient tile order and of the head streams
set -o pipefail
O=gpurun_out/r4a
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
python3 bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-200 $O/bench_default.json
for v in group_order ""; do
  MMSA_DISABLE=$v python3 bench.py --no-cpu-baseline --repeats 3 --exact-steps 0 > $O/ab_order_${v:-on}.json 2>> $O/ab.err
  python3 -c "import json,sys;d=json.loads(open('$O/ab_order_${v:-on}.json').read().strip().splitlines()[-1]);print('disable=[$v]',d['protocol']['ms_per_step_by_region'],d['roofline']['kernel_ms_per_step'],d['roofline']['isolated'])"
done
MMSA_HEAD_STREAMS=1 python3 bench.py --no-cpu-baseline --repeats 3 --exact-steps 0 > $O/ab_headstreams.json 2>> $O/ab.err
python3 -c "import json,sys;d=json.loads(open('$O/ab_headstreams.json').read().strip().splitlines()[-1]);print('head_streams',d['protocol']['ms_per_step_by_region'])"
MMSA_TWO_STREAMS=0 MMSA_WGRAD_STREAM=0 PROBE_B=16 timeout -k 10 240 python3 tools/microbench/graph_probe.py > $O/graph_single.log 2>&1; echo "graph single rc=$?"; tail -3 $O/graph_single.log
PROBE_B=16 timeout -k 10 120 python3 tools/microbench/graph_probe.py > $O/graph_guard.log 2>&1; echo "graph guarded rc=$?"; tail -2 $O/graph_guard.log
