#!/bin/bash
# Experiment builds of the library beside the shipped one: tools/microbench/build_variant.sh "<-DDEFINE ...>" <suffix>
#   -> multimodal_sentiment_aanalysis_amd/lib/libmmsa_hip_<suffix>.so, every source compiled with the defines (use with
#   MMSA_LIB=<path> on the GPU box; the in-tree library is untouched). -DMMSA_EXPERIMENTS turns the experiment hooks on
#   (README.md "Switches"): cost-model overrides, forced K splits, timing-only ablations, the 256 x 256 tile.
set -e
D=$1; S=$2
cd "$(dirname "$0")/../../multimodal_sentiment_aanalysis_amd/csrc"
make -j8 EXTRA="$D" BUILD=build_$S OUT=../lib/libmmsa_hip_$S.so
ls -la ../lib/libmmsa_hip_$S.so
