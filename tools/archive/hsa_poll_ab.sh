# host-side wake-up latency at the end of a timed block: HSA_ENABLE_INTERRUPT=0 (the runtime polls its completion signals) against the default (interrupts)
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['kernel_ms'])"; }
for rep in 1 2 3 4 5; do
  (unset HSA_ENABLE_INTERRUPT; run interrupts)
  (export HSA_ENABLE_INTERRUPT=0; run polling)
done
