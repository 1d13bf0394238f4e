cd $GRAFT_REPO_ROOT
for P in 0 1; do SFM_PAIR=$P WORKLOAD=cfg3_edge bash tools/pmc_quick.sh pair$P > gpurun_out/pmc_pair$P.txt 2>&1; done
tail -4 gpurun_out/pmc_pair0.txt gpurun_out/pmc_pair1.txt
for PT in "01,10" "00,00" "10,01" "11,11"; do echo "== SFM_PRIO_TABLE=$PT"; SFM_PRIO_TABLE=$PT timeout -k 10 120 python tools/pair_ab.py --rounds 4 --workloads cfg3_edge,cfg5 2>&1 | grep "main kernel"; done
for RL in "22,22,16,16" "26,22,16,16" "22,16,11,8" "19,16,16,8"; do echo "== SFM_CHUNK_ROWS_LIST=$RL"; SFM_CHUNK_ROWS_LIST=$RL timeout -k 10 120 python tools/pair_ab.py --rounds 4 --workloads cfg3_edge 2>&1 | grep "main kernel"; done
