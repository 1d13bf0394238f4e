#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py (the command the driver runs), then the HBM-traffic
# and SQ counters in their own --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; never combined with tracing).
# Raw CSVs land in gpurun_out/profiles_raw/<tag>/ ; tools/summarize_profiles.py condenses them into profiles/.
set -e
TAG=${1:-r02}
MODE=${2:-fused}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_raw/$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --mode $MODE"
PMC="python3 $R/bench.py --steps 6 --warmup 2 --min-time 0.001 --no-cpu-baseline --no-secondary --mode $MODE"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $PMC > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $PMC > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- $PMC > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_tcc -- $PMC > $OUT/pmc_tcc.log 2>&1 || true
tail -1 $OUT/trace.log
echo collected $OUT
