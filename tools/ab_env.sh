#!/bin/bash
# A/B of one environment toggle on the same box: tools/ab_env.sh VAR=VALUE [rounds]  (bench.py default steps, no CPU baseline)
V=$1; R=${2:-2}
for r in $(seq $R); do
  env $V python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V   ', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['achieved'])"
  python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['achieved'])"
done
