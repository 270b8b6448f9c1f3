set -e
python -m pytest tests/test_golden_gpu.py -q -x -k "a7 or trainer" > gpurun_out/ab_tests.log 2>&1 || { tail -20 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
for r in 1 2; do
  MMSA_G2_NOEPI=1 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('noepi', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['achieved'])"
  python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('epi  ', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['achieved'])"
done
python3 tools/microbench/stamp_check.py 2>&1 | grep -v amdgpu.ids
