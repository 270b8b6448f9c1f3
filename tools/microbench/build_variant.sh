#!/bin/bash
# A/B builds of the library: tools/microbench/build_variant.sh <-DDEFINE> <suffix>  ->  lib/libmmsa_hip_<suffix>.so with
# gemm_mfma2.hip compiled with the define (use with MMSA_LIB=<path> on the GPU box; the in-tree library is untouched)
set -e
D=$1; S=$2
cd "$(dirname "$0")/../../multimodal_sentiment_aanalysis_amd/csrc"
mkdir -p build_$S
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $D -c gemm_mfma2.hip -o build_$S/gemm_mfma2.o
OBJS=$(ls build/*.o | grep -v gemm_mfma2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmmsa_hip_$S.so $OBJS build_$S/gemm_mfma2.o
ls -la ../lib/libmmsa_hip_$S.so
