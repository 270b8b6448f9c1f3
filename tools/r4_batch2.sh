#!/bin/bash
# round-4 batch 2: full GPU tests; PMC traffic of the GEMM kernels with and without the grouped-launch tile order
set -o pipefail
O=gpurun_out/r4b
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
bash tools/run_pmc.sh > $O/pmc_on.log 2>&1; echo "pmc rc=$?"; cp gpurun_out/pmc_traffic.json $O/pmc_traffic_order_on.json
MMSA_DISABLE=group_order bash tools/run_pmc.sh > $O/pmc_off.log 2>&1; echo "pmc(off) rc=$?"; cp gpurun_out/pmc_traffic.json $O/pmc_traffic_order_off.json
python3 - <<'PY'
import json
for tag in ("on","off"):
    d=json.load(open(f"gpurun_out/r4b/pmc_traffic_order_{tag}.json"))
    k=d["kernels"].get("gemm2_kernel<4, 4, true, true, 0, false, 0>")
    print(tag, "grouped BERT wgrad:", k, "all:", d["all_gemm"])
PY
