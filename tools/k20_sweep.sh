NVF_BENCH_DEBUG=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-sweep --no-cpu-baseline --no-pmc --no-epoch --no-direct --sustained-s 0 2>&1 >/dev/null | grep "\[bench\] region"
NVF_BENCH_DEBUG=1 timeout -k 10 200 python bench.py --steps 400 --warmup 40 --no-sweep --no-cpu-baseline --no-pmc --no-epoch --no-direct --sustained-s 0 2>&1 >/dev/null | grep "\[bench\] region"
