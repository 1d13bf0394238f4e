# does the step time depend on the NUMA node the bench process runs on?  (tools/numa_ab.sh > gpurun_out/numa_ab.txt)
ls -la /dev/dri/ 2>&1 | head -20
echo "ROCR_VISIBLE_DEVICES=$ROCR_VISIBLE_DEVICES HIP_VISIBLE_DEVICES=$HIP_VISIBLE_DEVICES"
for rep in 1 2 3; do for cpus in 0-63 64-127; do
taskset -c $cpus timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cpus $cpus', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
