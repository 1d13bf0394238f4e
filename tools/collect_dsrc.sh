set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_loss_gpu.py tests/test_loss_edges_gpu.py tests/test_api_gpu.py -x -q -m gpu -k "d_src or reference_order or src" > gpurun_out/dsrc_tests.log 2>&1 || exit 1
tail -1 gpurun_out/dsrc_tests.log
timeout -k 10 300 python tools/dsrc_time.py cfg3_edge cfg3_smooth_disp cfg2 cfg5_2src ref_b4 > gpurun_out/r06_dsrc_time.txt 2>&1
for wl in cfg3_edge cfg3_smooth_disp; do SFM_DSRC_COUNT=1 timeout -k 10 100 python tools/dsrc_once.py $wl 6 2>&1 | grep "dsrc counters" | sed "s/^/$wl: /" >> gpurun_out/r06_dsrc_time.txt; done
bash tools/pmc_dsrc.sh final > gpurun_out/r06_pmc_dsrc.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dsrc_kt -- python3 $GRAFT_REPO_ROOT/tools/dsrc_once.py cfg3_edge 40 > $GRAFT_REPO_ROOT/gpurun_out/dsrc_kt.log 2>&1
cd $GRAFT_REPO_ROOT; cp $(ls -t gpurun_out/dsrc_kt/*/*kernel_stats.csv | head -1) gpurun_out/r06_kernel_stats_cfg3_edge_d_src.csv
echo done
