for i in 1 2; do timeout -k 10 100 python bench.py --steps 30 --warmup 10 --no-extras 2>/dev/null | cut -c1-140 || exit 1; done
