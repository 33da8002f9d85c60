for n in 1000 0 1000 0; do echo defer=$n; MM_DEFER_DW_LAYERS=$n timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-extras | cut -c1-150 || exit 1; done
