#!/bin/bash
# Rank-weighted chunk plans (SFM_CLASS_PLAN) x issue-priority tables (SFM_PRIO_TABLE) on one workload: short bench.py runs, one line each.
# usage (GPU box): WORKLOAD=cfg3_edge tools/class_sweep.sh > gpurun_out/class_sweep.txt
S3="1x8+1x8"
PLANS=(
  ""                                              # the planner's own choice
  "10x13;5x13;2x16;$S3"                           # 3072 equal items
  "3x14+3x13+4x12;2x14+2x13+1x10;1x16+1x16;$S3"
  "3x15+3x13+4x11;2x15+2x13+1x8;1x17+1x15;$S3"
  "3x16+3x12+4x11;2x16+2x12+1x8;1x18+1x14;$S3"
  "3x17+3x13+4x10;2x17+2x13+1x4;1x18+1x14;$S3"
  "3x18+3x12+4x10;2x18+2x12+1x4;1x19+1x13;$S3"
  "3x19+3x11+4x10;2x19+2x11+1x4;1x20+1x12;$S3"
)
PRIOS=(${PRIOS:-"" "210,210" "012,012" "000,000"})
for pr in "${PRIOS[@]}"; do
  for pl in "${PLANS[@]}"; do
    envs=""
    [ -n "$pl" ] && envs="$envs SFM_CLASS_PLAN=$pl"
    [ -n "$pr" ] && envs="$envs SFM_PRIO_TABLE=$pr"
    out=$(env $envs timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --mode fused --workload ${WORKLOAD:-cfg3_edge} 2>/dev/null | tail -1)
    echo "[prio=${pr:-default} plan=${pl:-default}] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_step=%.4f"%d["ms_per_step"], "main_us=%.2f"%(d["roofline"]["kernel_ms"]*1e3), "frac=%.4f"%d["roofline"]["frac"])')"
  done
done
