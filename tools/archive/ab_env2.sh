#!/bin/bash
# usage: ab_env2.sh "<bench args>" "ENV..." ...
ARGS="$1"; shift
for envs in "$@"; do
  out=$(env $envs timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary $ARGS 2>/dev/null | tail -1)
  echo "[$ARGS] [$envs] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_step=%.4f"%d["ms_per_step"], "main_us=%.2f"%(d["roofline"]["kernel_ms"]*1e3), "value=%.0f"%d["value"])')"
done
