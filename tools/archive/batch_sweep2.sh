#!/bin/bash
# kernel time vs batch at FIXED chunk height: how much of a launch is one wave's own latency, how much is contention
for ch in ${CHS:-15}; do
for b in 8 16 24 32 40 48 64; do
    out=$(SFM_CHUNK_ROWS=$ch timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --mode fused --batch $b 2>/dev/null | tail -1)
    echo "rows=$ch B=$b: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_step=%.4f"%d["ms_per_step"], "main_us=%.2f"%(d["roofline"]["kernel_ms"]*1e3), "value=%.0f"%d["value"])')"
done
done
