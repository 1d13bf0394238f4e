# per-process modes of the step time with and without address-space randomisation (setarch -R), eight processes each, interleaved
run() { "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'])"; }
cat /proc/sys/kernel/randomize_va_space
for rep in 1 2 3 4 5 6 7 8; do
  echo -n "aslr    "; run
  echo -n "no-aslr "; run setarch -R
done
