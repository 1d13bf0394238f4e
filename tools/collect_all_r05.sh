set -x
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 tools/op_cost.hip -o /tmp/op_cost && /tmp/op_cost > gpurun_out/r05_op_cost_microbench.txt 2>&1
for w in cfg3_edge cfg3 cfg2 cfg5 cfg5_2src ref_b4; do bash tools/collect_profiles.sh r05 $w hwc fused; done
bash tools/collect_profiles.sh r05 cfg3_edge planar fused
bash tools/collect_profiles.sh r05 cfg3_edge hwc separate
cd $GRAFT_REPO_ROOT
SFM_TRACE_SMOOTH=edge_aware SFMWARP_LIB=sfm-learner-chainer_amd/libsfmwarp_stamps.so python tools/trace_waves.py fused > gpurun_out/r05_wave_stage_stamps.txt 2>&1
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_args.json 2> gpurun_out/r05_bench_driver_args.err
echo collection done
