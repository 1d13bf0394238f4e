# One round's whole collection on the GPU box, in two gpurun calls (a call is limited to 20 minutes):
#   gpurun --timeout 1200 -- 'bash tools/collect_all.sh r06 1'     op costs, the headline variant + layout / mode variants, stage stamps, the two bench lines
#   gpurun --timeout 1200 -- 'bash tools/collect_all.sh r06 2'     the other workloads' variants
# then here: summarize_profiles.py, issue_model.py, perf_table.py, parity_table.py --design, gen_design.py (tools/README.md).
set -x
TAG=${1:-r06}
PART=${2:-1}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$PART" = "1" ]; then
  hipcc --offload-arch=gfx950 -O3 tools/op_cost.hip -o /tmp/op_cost && /tmp/op_cost > gpurun_out/${TAG}_op_cost_microbench.txt 2>&1
  bash tools/collect_profiles.sh $TAG cfg3_edge hwc fused
  bash tools/collect_profiles.sh $TAG cfg3_edge planar fused
  bash tools/collect_profiles.sh $TAG cfg3_edge hwc separate
  cd $GRAFT_REPO_ROOT
  SFM_TRACE_SMOOTH=edge_aware SFMWARP_LIB=sfm-learner-chainer_amd/libsfmwarp_stamps.so python tools/trace_waves.py fused > gpurun_out/${TAG}_wave_stage_stamps.txt 2>&1
  python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_args.json 2> gpurun_out/${TAG}_bench_driver_args.err
else
  for w in cfg3 cfg2 cfg5 cfg5_2src ref_b4; do bash tools/collect_profiles.sh $TAG $w hwc fused; done
fi
echo collection part $PART done
