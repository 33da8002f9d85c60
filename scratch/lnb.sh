for cap in 256 512; do echo cap $cap; MM_LN_CAP=$cap timeout -k 10 100 python benchmarks/kernel_bench.py --what ln 2>&1 | grep -E '"rows": (10080|80640), "C": 1600' | cut -c 40-200 || exit 1; done
