echo "B=32 default"; timeout -k 10 100 python bench.py --steps 30 --warmup 10 --no-extras 2>/dev/null | cut -c1-140 || exit 1
echo "B=32 DP env"; TENSILE_STREAMK_DATA_PARALLEL=1 timeout -k 10 100 python bench.py --steps 30 --warmup 10 --no-extras 2>/dev/null | cut -c1-140 || exit 1
echo "B=52 DP env, deferral forced"; TENSILE_STREAMK_DATA_PARALLEL=1 MM_DEFER_MAX_ROWS=100000000 timeout -k 10 60 python bench.py --batch 52 --steps 20 --warmup 5 --no-extras 2>/dev/null | cut -c1-140 || { echo HUNG52; exit 1; }
echo "B=256 DP env, deferral forced"; TENSILE_STREAMK_DATA_PARALLEL=1 MM_DEFER_MAX_ROWS=100000000 timeout -k 10 90 python bench.py --batch 256 --steps 4 --warmup 2 --no-extras 2>/dev/null | cut -c1-140 || { echo HUNG256; exit 1; }
