#!/bin/bash
# Lab evidence behind DESIGN.md 4 (K2b main-loop study): ceiling probes, cycle stamps of the adopted loop and of the experiments
set -o pipefail
O=gpurun_out/r2/lab; mkdir -p $O
{
echo "### benchmarks/mfma_peak.hip"; timeout -k 10 120 benchmarks/bin/mfma_peak || exit 1
echo; echo "### benchmarks/dma_issue.hip"; timeout -k 10 120 benchmarks/bin/dma_issue || exit 1
echo; echo "### gemm_lab 224x320 tile (stamps: early waves s1->s2 barrier, s2->s3 DMA issue, s3->s4 first half step, s4->s5 second; late waves: s2->s3 held MFMAs, s3->s4 DMA, s4->s5 first half step, s5->s6 reads of the second)"
for sh in "10080 1600 6400" "10080 1600 1600"; do timeout -k 10 60 benchmarks/bin/gemm_lab_75 $sh || exit 1; done
echo; echo "### gemm_lab 256x256 tile, main-loop variants (MM_GEMM_BIG_LOOP)"
for l in two pipe ring reg; do echo "== $l"; MM_GEMM_BIG_LOOP=$l timeout -k 10 60 benchmarks/bin/gemm_lab_84 8192 8192 8192 || exit 1; done
echo; echo "### weight-gradient tile 256x256 (tn)"
timeout -k 10 60 benchmarks/bin/gemm_lab_84 6400 1600 10080 tn || exit 1
} > $O/r02_gemm_lab.txt 2>&1
tail -5 $O/r02_gemm_lab.txt
