SH="32,16,100,315,221;32,8,128,238,159"
for v in base a b c; do echo $v; L=""; [ $v != base ] && L=$PWD/multimodal_amd/lib/libmmhip_$v.so; MM_LIB=$L timeout -k 10 200 python benchmarks/kernel_bench.py --what attnx --shapes "$SH" 2>&1 | grep prefix | cut -c 24-60,130-300 || exit 1; done
