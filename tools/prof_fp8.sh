#!/bin/bash
# rocprofv3 kernel trace of the B = 128 step in fp8 and bf16 mode on the GPU box: tools/prof_fp8.sh <prefix>
P=${1:-q}
R=$GRAFT_REPO_ROOT
for prec in fp8 bf16; do
  (cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/${P}_${prec}_prof && MMSA_BENCH_NOPROF=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${P}_${prec}_prof -o run -- python3 $R/bench.py --batch 128 --precision $prec --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline > $R/gpurun_out/${P}_${prec}_prof.log 2>&1) || exit 1
  python3 tools/prof_summary.py gpurun_out/${P}_${prec}_prof/run_results.db 8 > gpurun_out/${P}_${prec}_prof.md
  head -40 gpurun_out/${P}_${prec}_prof.md | cut -c1-150
done
