# per-process levels of the step time with and without the input warm read of bench.py (--no-input-warm), 12 + 12 processes interleaved
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline $2 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['kernel_ms'])"; }
for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do
  run set-up-state --no-input-warm
  run inputs-read-once ""
done
