#!/bin/bash
# usage: tools/ab_env.sh "ENV1=.. ENV2=.." "ENV..." ... ; prints fused kernel ms for each environment setting
for envs in "$@"; do
  out=$(env $envs timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --mode ${MODE:-fused} --workload ${WORKLOAD:-cfg3} 2>/tmp/ab_env_err.txt | tail -1)
  echo "[$envs] $(grep balance /tmp/ab_env_err.txt | head -1) $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_step=%.4f"%d["ms_per_step"], "main_us=%.2f"%(d["roofline"]["kernel_ms"]*1e3), "value=%.0f"%d["value"])')"
done
