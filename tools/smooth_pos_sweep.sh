#!/bin/bash
for p in 020 000 222 002 200 022 220 202 010 012 210 011 110 101 121 021 120 212; do
  out=$(SFM_SMOOTH_POS=$p timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --workload ${WORKLOAD:-cfg3_edge} 2>/dev/null | tail -1)
  echo "[pos=$p] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_step=%.4f"%d["ms_per_step"], "main_us=%.2f"%(d["roofline"]["kernel_ms"]*1e3))')"
done
