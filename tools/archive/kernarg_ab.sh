# per-process modes of the step time: HIP_FORCE_DEV_KERNARG unset / 1 / 0, eight processes each, interleaved (one box)
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['kernel_ms'])"; }
for rep in 1 2 3 4 5 6 7 8; do
  (unset HIP_FORCE_DEV_KERNARG; run unset)
  (export HIP_FORCE_DEV_KERNARG=1; run dev=1)
  (export HIP_FORCE_DEV_KERNARG=0; run dev=0)
done
