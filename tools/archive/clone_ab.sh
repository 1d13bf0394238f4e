# per-process levels of the step time with the bench's pyramids copied once into fresh arrays right after they are made (SFM_BENCH_CLONE_INPUTS=t / ts) and without
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['kernel_ms'])"; }
for rep in 1 2 3 4 5 6 7 8 9 10; do
  (unset SFM_BENCH_CLONE_INPUTS; run as-made)
  (export SFM_BENCH_CLONE_INPUTS=t; run tgt-cloned)
  (export SFM_BENCH_CLONE_INPUTS=ts; run tgt+src-cloned)
done
