#!/bin/bash
# build_lab.sh TM TN  ->  benchmarks/bin/gemm_lab_<TM><TN>
cd "$(dirname "$0")/.." && mkdir -p benchmarks/bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-function -DGEMM_LAB -DGEMM_LAB_TM=$1 -DGEMM_LAB_TN=$2 -DGEMM_LAB_KT0=${3:-16} $EXTRA -I multimodal_amd/csrc -I include -o benchmarks/bin/gemm_lab_$1$2 benchmarks/gemm_lab.hip 2>&1 | grep -E "error|warning: var" | head
