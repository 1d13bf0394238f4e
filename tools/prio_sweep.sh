#!/bin/bash
# usage (GPU box): tools/prio_sweep.sh "012,210" "013,310" ...  -- main-kernel time of the fused cfg3 launch per issue-priority table
P=sfm-learner-chainer_amd
for t in "$@"; do
  echo "$t $(SFM_PRIO_TABLE=$t timeout -k 10 120 python tools/ab_inproc.py --workload ${WORKLOAD:-cfg3_edge} --rounds 7 --iters 30 $P/libsfmwarp.so 2>&1 | tail -1 | sed 's/.*main kernel us: //')"
done
