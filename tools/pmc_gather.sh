#!/bin/bash
# Vector-memory-pipeline counters of the main kernel on the rough (default) and the smooth disparity field, 128x416 and 256x832:
# texture-addresser busy cycles and L1 (TCP) accesses per launch -- the evidence behind "the rough field costs line look-ups per gather"
# (profiles/r05_process_modes.txt 3.).  Counters only, no tracing.   tools/pmc_gather.sh ; python tools/pmc_gather_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_gather
rm -rf $OUT; mkdir -p $OUT
for WL in cfg3_edge cfg3_smooth_disp cfg5_2src cfg5_2src_smooth_disp; do
  CMD="python3 $R/bench.py --steps 6 --warmup 2 --min-time 0.001 --no-cpu-baseline --no-secondary --workload $WL"
  rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $OUT/${WL}_p1 -- $CMD > $OUT/${WL}_p1.log 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/${WL}_p2 -- $CMD > $OUT/${WL}_p2.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/${WL}_p3 -- $CMD > $OUT/${WL}_p3.log 2>&1
  echo $WL done
done
