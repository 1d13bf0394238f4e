#!/bin/bash
# usage: tools/prio_sweep.sh "012,210" "102,210" ... : fused main-kernel time for issue-priority tables (SFM_PRIO_TABLE), 200-step runs, 2 rounds
for r in 1 2; do
for t in "$@"; do
  out=$(SFM_PRIO_TABLE=$t timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$t $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_step=%.4f"%d["ms_per_step"], "main_us=%.2f" % (d["kernel_ms"]["fused_main"]*1e3))')"
done
done
