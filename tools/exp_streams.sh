#!/bin/bash
# A/B on one box: the two encoders on two HIP streams (FusedTrainStep(two_streams=True)) x a cap on the persistent GEMM grid
# (MMSA_G2_CUS): does leaving CUs free let one encoder's small / HBM-bound kernels run beside the other's GEMMs?
for cus in 0 248 240 224; do for ts in 0 1; do
  v=""; [ $cus -gt 0 ] && v="MMSA_G2_CUS=$cus"
  line=$(env $v MMSA_TWO_STREAMS=$ts MMSA_BENCH_NOPROF=1 python3 bench.py --steps 40 --warmup 10 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "G2_CUS=$cus two_streams=$ts $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["protocol"]["ms_per_step_by_region"], d["value"])')"
done; done
