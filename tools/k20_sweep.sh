run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-sweep --no-cpu-baseline --no-pmc --no-epoch --no-direct --sustained-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['repeats']['ms_per_step'])"; }
run spin; NVF_BENCH_SPIN=0 run nospin; run spin; NVF_BENCH_SPIN=0 run nospin
