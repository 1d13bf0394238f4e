# per-process modes of the step time with every array carved out of ONE pre-reserved block (SFM_BENCH_ARENA_GB) and without, 12 + 12 processes
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['kernel_ms'])"; }
for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do
  (unset SFM_BENCH_ARENA_GB; run separate)
  (export SFM_BENCH_ARENA_GB=4; run arena)
done
