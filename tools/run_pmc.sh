# two separate --pmc passes (FETCH_SIZE / WRITE_SIZE do not fit one pass), kernel-trace only (no other trace domains)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$c
  MMSA_BENCH_NOPROF=1 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1 || exit 1
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/run_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/run_counter_collection.csv gpurun_out/pmc_traffic.json
