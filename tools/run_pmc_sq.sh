#!/bin/bash
# One SQ/GRBM --pmc pass over the default step (kernel-trace only): matrix-pipe busy cycles, LDS bank conflicts, wait buckets.
# Run on the GPU box; writes gpurun_out/pmc_sq.json (tools/pmc_sq.py). SURVEY §8(d): "MFMA busy %, LDS bank conflicts".
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_SQ
MMSA_BENCH_NOPROF=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_SQ -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_SQ.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/pmc_SQ.log; exit 1; }
cd $GRAFT_REPO_ROOT && python3 tools/pmc_sq.py gpurun_out/pmc_SQ/run_counter_collection.csv gpurun_out/pmc_sq.json
