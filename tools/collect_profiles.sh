#!/bin/bash
# Runs on the GPU box (via gpurun): for ONE variant (workload, layout, mode) of bench.py -- the command the driver runs --
# rocprofv3 kernel trace + stats, then the HBM-traffic and SQ counters in their own --pmc passes (FETCH_SIZE and WRITE_SIZE do not
# fit one pass on gfx950; counters are never combined with tracing).  Raw CSVs land in gpurun_out/profiles_raw/<tag>/<variant>/ ;
# tools/summarize_profiles.py <tag> condenses every variant found there into profiles/<tag>_*.
#   usage: tools/collect_profiles.sh <tag> [workload=cfg3_edge] [layout=hwc] [mode=fused]
#   e.g.   for w in cfg3_edge cfg3 cfg2 cfg5 cfg5_2src; do tools/collect_profiles.sh r04 $w; done; tools/collect_profiles.sh r04 cfg3_edge planar
set -e
TAG=${1:-r05}
WL=${2:-cfg3_edge}
LAYOUT=${3:-hwc}
MODE=${4:-fused}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_raw/$TAG/${WL}_${LAYOUT}_${MODE}
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-secondary --mode $MODE --workload $WL --layout $LAYOUT"
CMD="python3 $R/bench.py --steps 20 --warmup 5 $ARGS"
PMC="python3 $R/bench.py --steps 6 --warmup 2 --min-time 0.001 $ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $PMC > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $PMC > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- $PMC > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_tcc -- $PMC > $OUT/pmc_tcc.log 2>&1 || true
tail -1 $OUT/trace.log | cut -c1-300
echo collected $OUT
