SH="10080,6400,1600;10080,4800,1600;10080,1600,1600;10080,1600,4800;10080,1600,6400;4096,4096,4096;8192,8192,8192"
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "gemm" 2>&1 | tail -3 || exit 1
echo four-stage; timeout -k 10 200 python benchmarks/kernel_bench.py --what gemmx --shapes "$SH" 2>&1 | grep bfloat16 | cut -c 18-60,100-200 || exit 1
echo two-stage; MM_GEMM_BIG_2STAGE=1 timeout -k 10 200 python benchmarks/kernel_bench.py --what gemmx --shapes "$SH" 2>&1 | grep bfloat16 | cut -c 18-60,100-200
