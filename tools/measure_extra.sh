#!/bin/bash
# The non-default bench lines of a round on the GPU box: tools/measure_extra.sh <prefix>  -> gpurun_out/<prefix>_extra.jsonl, <prefix>_driverlike.json
P=${1:-x}
O=gpurun_out/${P}_extra.jsonl
: > $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${P}_driverlike.json 2> gpurun_out/${P}_extra.err || exit 1
for args in "--precision fp32 --steps 30 --warmup 8 --repeats 1" "--model large --steps 30 --warmup 8 --repeats 1" \
            "--batch 128 --steps 30 --warmup 8 --repeats 1" "--batch 128 --precision fp8 --steps 30 --warmup 8 --repeats 1" \
            "--mode fwd --steps 30 --warmup 8 --repeats 1" "--host-inputs --steps 30 --warmup 8 --repeats 1"; do
  python3 bench.py $args --no-cpu-baseline >> $O 2>> gpurun_out/${P}_extra.err || exit 1
  tail -1 $O | cut -c1-200
done
