timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1
for n in 1 0 1 0; do echo encside $n; MM_ENC_SIDE=$n timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras | cut -c1-150 || exit 1; done
