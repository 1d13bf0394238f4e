#!/bin/bash
# usage (on the GPU box): tools/pmc_dsrc.sh <tag> [workload]  -- SQ counters of the two launches of a step with d_src bound, one rocprofv3 --pmc pass per group
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
OUT=$R/gpurun_out/pmc_dsrc_$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python3 $R/tools/dsrc_once.py ${2:-cfg3_edge} 6"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            if "dsrc" in k or "loss_kernel" in k:
                print(d.split("/")[-2], k, {c: round(sum(x) / len(x), 1) for c, x in v.items()})
PY
