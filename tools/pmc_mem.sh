#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_mem
rm -rf $OUT; mkdir -p $OUT
CMD="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --mode fused"
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1
echo done
