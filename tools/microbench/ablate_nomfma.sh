#!/bin/bash
# Builds lib/libmmsa_hip_nomfma.so = the library with the persistent GEMM's matrix instructions compiled out (results are
# wrong; everything else - LDS-DMA, counted waits, barriers, fragment reads, epilogue - unchanged). Run here (CPU box), then
# on the GPU box:  MMSA_LIB=multimodal_sentiment_aanalysis_amd/lib/libmmsa_hip_nomfma.so python tools/microbench/bench_vendor_gemm.py
set -e
cd "$(dirname "$0")/../../multimodal_sentiment_aanalysis_amd/csrc"
mkdir -p build_nomfma
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DG2_ABLATE_NO_MFMA -c gemm_mfma2.hip -o build_nomfma/gemm_mfma2.o
OBJS=$(ls build/*.o | grep -v gemm_mfma2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmmsa_hip_nomfma.so $OBJS build_nomfma/gemm_mfma2.o
ls -la ../lib/libmmsa_hip_nomfma.so
