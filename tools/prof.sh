#!/bin/bash
# rocprofv3 kernel trace + a PMC pass of bench.py; summaries land in gpurun_out/prof_*
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:-fused}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --mode $MODE > $R/gpurun_out/prof_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/prof_pmc1 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --mode $MODE > $R/gpurun_out/prof_pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_pmc2 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --mode $MODE > $R/gpurun_out/prof_pmc2.log 2>&1
echo done
