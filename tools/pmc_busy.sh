#!/bin/bash
# Derived unit-busy metrics of the fused launch, one rocprofv3 --pmc pass per group (counters only, no tracing)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_busy
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
CMD="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --mode ${MODE:-fused}"
i=0
for grp in "VALUBusy SALUBusy" "MemUnitBusy MemUnitStalled" "VALUUtilization LDSBankConflict" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAVES" "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            if "loss_kernel" in k:
                print(d.split("/")[-2], k, {c: round(sum(x) / len(x), 1) for c, x in v.items()})
PY
