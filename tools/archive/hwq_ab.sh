# per-process modes of the step time against the number of hardware queues ROCclr creates (GPU_MAX_HW_QUEUES), 10 + 10 + 10 processes
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['kernel_ms'])"; }
for rep in 1 2 3 4 5 6 7 8 9 10; do
  (unset GPU_MAX_HW_QUEUES; run default)
  (export GPU_MAX_HW_QUEUES=1; run hwq=1)
  (export GPU_MAX_HW_QUEUES=2; run hwq=2)
done
