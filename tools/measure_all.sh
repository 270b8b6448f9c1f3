#!/bin/bash
# The round's measurement set on the GPU box: tools/measure_all.sh <prefix>  (then, back home: tools/publish_profiles.sh <prefix>)
set -e -o pipefail
P=${1:-f}
# a red test run stops the measurement set: published numbers always come from a build whose GPU tests pass
python -m pytest tests -m gpu -q > gpurun_out/${P}_tests.log 2>&1 || { tail -30 gpurun_out/${P}_tests.log; echo "TESTS FAILED: no measurements taken"; exit 1; }
tail -1 gpurun_out/${P}_tests.log
MMSA_PROF_DUMP=gpurun_out/${P}_shapes.csv python3 bench.py > gpurun_out/${P}_bench.json 2> gpurun_out/${P}_bench.err
cut -c1-160 gpurun_out/${P}_bench.json
R=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/${P}_prof && MMSA_BENCH_NOPROF=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${P}_prof -o run -- python3 $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline > $R/gpurun_out/${P}_prof.log 2>&1)
echo "rocprof done"
bash tools/run_pmc.sh
echo "pmc done"
bash tools/run_pmc_sq.sh > gpurun_out/${P}_pmc_sq.log 2>&1 || { tail -5 gpurun_out/${P}_pmc_sq.log; exit 1; }
echo "pmc sq done"
# rehearsal of the multi-rank path on the one GPU of the box (gloo moves the gradients; both ranks share cuda:0)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --repeats 1 --backend gloo --no-cpu-baseline > gpurun_out/${P}_ddp2.log 2>&1 || { tail -20 gpurun_out/${P}_ddp2.log; exit 1; }
tail -1 gpurun_out/${P}_ddp2.log | cut -c1-200
