SH="10080,6400,1600;10080,1600,6400;4096,4096,4096;8192,8192,8192"
echo 4wave-pipe; MM_GEMM_BIG_4WAVE=1 timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "gemm_large" 2>&1 | tail -1 || exit 1
MM_GEMM_BIG_4WAVE=1 timeout -k 10 200 python benchmarks/kernel_bench.py --what gemmx --shapes "$SH" 2>&1 | grep bfloat16 | cut -c 18-60,100-200 || exit 1
echo 8wave-pipe; MM_GEMM_BIG_PIPE=1 timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "gemm_large" 2>&1 | tail -1 || exit 1
MM_GEMM_BIG_PIPE=1 timeout -k 10 200 python benchmarks/kernel_bench.py --what gemmx --shapes "$SH" 2>&1 | grep bfloat16 | cut -c 18-60,100-200 || exit 1
echo 8wave; timeout -k 10 200 python benchmarks/kernel_bench.py --what gemmx --shapes "$SH" 2>&1 | grep bfloat16 | cut -c 18-60,100-200
