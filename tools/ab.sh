#!/bin/bash
# usage: tools_ab.sh lib1 lib2 ... ; prints kernel ms for fused and separate modes at chunk rows 8/16
for lib in "$@"; do
  for ch in ${CHS:-0 16}; do
    for mode in separate fused; do
      out=$(SFMWARP_LIB=$PWD/sfm-learner-chainer_amd/$lib SFM_CHUNK_ROWS=$ch timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --mode $mode 2>/dev/null | tail -1)
      echo "$lib ch=$ch $mode: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_step=%.4f"%d["ms_per_step"], d["kernel_ms"], "value=%.0f"%d["value"])')"
    done
  done
done
