#!/bin/bash
# Short rocprofv3 kernel trace of the default step on the GPU box: tools/prof_step.sh <prefix> [grep pattern]
# prints the per-kernel table rows matching the pattern (default: finalize / reduce kernels)
P=${1:-p}; PAT=${2:-finalize|reduce|tiny|colsum}
R=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/${P}_prof && MMSA_BENCH_NOPROF=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${P}_prof -o run -- python3 $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline > $R/gpurun_out/${P}_prof.log 2>&1)
python3 tools/prof_summary.py gpurun_out/${P}_prof/run_results.db 13 > gpurun_out/${P}_prof.md
head -3 gpurun_out/${P}_prof.md
grep -E "$PAT" gpurun_out/${P}_prof.md | cut -c1-160
