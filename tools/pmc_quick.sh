#!/bin/bash
# usage (on the GPU box): tools/pmc_quick.sh <tag> [lib.so]  -- SQ counters of the fused cfg3 launch, one rocprofv3 --pmc pass per group
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
[ -n "$2" ] && export SFMWARP_LIB=$R/sfm-learner-chainer_amd/$2
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --min-time 0.001 --workload ${WORKLOAD:-cfg3}"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            if "loss_kernel" in k:
                print(d.split("/")[-2], k, {c: round(sum(x) / len(x), 1) for c, x in v.items()})
PY
