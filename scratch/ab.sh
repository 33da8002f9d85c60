timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1
for n in 0 1 0; do if [ $n = 1 ]; then export MM_OPT_OVERLAP=1; else unset MM_OPT_OVERLAP; fi; echo overlap=$n; timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-extras | cut -c1-150 || exit 1; done
