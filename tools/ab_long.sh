#!/bin/bash
# usage: tools/ab_long.sh libA.so libB.so [rounds]: alternating 200-step fused runs, prints the main-kernel time of each
for r in $(seq 1 ${3:-5}); do
  for lib in "$1" "$2"; do
    out=$(SFMWARP_LIB=$PWD/sfm-learner-chainer_amd/$lib timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$lib $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_step=%.4f"%d["ms_per_step"], "main_us=%.2f" % (d["kernel_ms"]["fused_main"]*1e3), "other_mode_ms=%.4f" % d["other_mode"]["ms_per_step"])')"
  done
done
