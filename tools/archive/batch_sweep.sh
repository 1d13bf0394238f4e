#!/bin/bash
for b in 4 8 11 16 22 32 48 64; do
  for mode in fused; do
    out=$(timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --mode $mode --batch $b 2>/dev/null | tail -1)
    echo "B=$b $mode: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_step=%.4f"%d["ms_per_step"], "main_us=%.2f"%(d["roofline"]["kernel_ms"]*1e3), "value=%.0f"%d["value"])')"
  done
done
