run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --prime-rounds $1 --no-sweep --no-cpu-baseline --no-pmc --no-epoch --no-direct --sustained-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rounds $1', d['value'], d['repeats']['ms_per_step'], d['config']['primed_steps'])"; }
run 1; run 30; run 100; run 1; run 30; run 100
