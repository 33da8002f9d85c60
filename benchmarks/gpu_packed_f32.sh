#!/bin/bash
# Repeatability of the optimizer update beside other kernels, with the reduce pass's sums paired into v_pk_fma_f32
# (benchmarks/bin/libmmhip_packed.so = `bash benchmarks/build_variant.sh packed elementwise ""`, built beforehand) and with the
# product build (scalar fmas).  Output: one line per (build, neighbour): how many of 8 updates differed from the update run alone.
set -o pipefail
O=gpurun_out/r3/ev
mkdir -p $O
out=$O/r03_packed_f32_repeatability.txt
echo "# optimizer update of the 8 decoder layers (248 M parameters) on the side stream, inputs restored before every run;" > $out
echo "# neighbour = what the main stream runs meanwhile; perturbed = runs (of 8) whose parameters differ from the update run alone" >> $out
for lib in benchmarks/bin/libmmhip_packed.so ""; do
  for b in gemmk gemmbig encoder mm ln k1 add; do
    n=$(MM_LIB=$lib BESIDE=$b REPS=8 timeout -k 10 300 python benchmarks/opt_race_check.py 2>/dev/null | grep "tensors differing" | grep -v alone | grep -vc "differing: 0 ") || true
    echo "build=$([ -n "$lib" ] && echo packed-f32 || echo product) neighbour=$b perturbed=$n/8" | tee -a $out
  done
done
