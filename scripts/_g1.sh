timeout 300 python bench_ljpeg.py --only cfg4small --no-cpu 2>&1 | tail -32
timeout 300 python bench_ljpeg.py --only cfg4small1 --no-cpu 2>&1 | grep -A12 "ms_per_step"
