#!/usr/bin/env python3
"""Stand-alone timing of the operator-level kernels (the reference-named functions of sfm_ops.hip) at the cfg3 shape:
B=32, 128x416, one source.  Prints per operator the mean launch time and the algorithmic GB/s it corresponds to."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "sfm-learner-chainer_amd"
ops = importlib.import_module(PKG + ".ops"); synth = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda:0")
B, H, W = 32, 128, 416
P = H * W
d = synth.make_inputs(B=B, H=H, W=W, n_src=2, n_scales=4, seed=1)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
src = t(d["src_pyr"][0][:, :3])
tgt = t(d["tgt_pyr"][0])
disp = t(d["disps"][0])
depth3 = (1.0 / disp).expand(B, 3, H, W).reshape(B, 3, P).contiguous()
pose = t(d["poses"][0]); K = t(d["intrinsics"][:, 0])
gy = torch.randn((B, 3, H, W), device=dev)
grid = torch.rand((B, 2, H, W), device=dev) * 2 - 1
gridp = torch.stack([torch.rand((B, H, W), device=dev) * (W - 1), torch.rand((B, H, W), device=dev) * (H - 1)], 1).contiguous()
# a warp field, the grids these samplers see in the path: the identity lattice plus a smooth displacement of a few pixels
yy, xx = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
flow_u = 3.0 * torch.sin(yy / 17.0) + 2.0 * torch.cos(xx / 23.0) + 0.37
flow_v = 2.0 * torch.cos(yy / 13.0) - 1.5 * torch.sin(xx / 29.0) + 0.61
gridp_warp = torch.stack([xx + flow_u, yy + flow_v], 0).expand(B, 2, H, W).contiguous()                    # pixel units (interp sampler)
grid_warp = torch.stack([gridp_warp[:, 0] / (W - 1) * 2 - 1, gridp_warp[:, 1] / (H - 1) * 2 - 1], 1).contiguous()   # [-1, 1]
full_src = t(d["src_pyr"][0])
logits = [torch.randn((B, 1, H >> s, W >> s), device=dev) for s in range(4)]
dd = ops.disp_act_fwd(logits)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


px = B * P
rows = [
    ("pose_proj_fwd (B,6)->(B,4,4)", lambda: ops.pose_proj_fwd(pose, K), None),
    ("warp_fwd  projective_inverse_warp", lambda: ops.warp_fwd(src, depth3, pose, K), px * (12 + 12 + 12)),        # src + 3 depth rows + warped
    ("warp_bwd  (d_depth 3 rows, d_pose)", lambda: ops.warp_bwd(src, depth3, pose, K, gy), px * (12 + 12 + 12 + 12)),
    ("sampler_fwd  F.spatial_transformer_sampler, warp field", lambda: ops.sampler_fwd(src, grid_warp), px * (12 + 8 + 12)),
    ("sampler_bwd  (gx, ggrid), warp field", lambda: ops.sampler_bwd(src, grid_warp, gy), px * (12 + 8 + 12 + 12 + 8)),
    ("sampler_fwd  uniformly random grid (worst case)", lambda: ops.sampler_fwd(src, grid), px * (12 + 8 + 12)),
    ("sampler_bwd  uniformly random grid (worst case)", lambda: ops.sampler_bwd(src, grid, gy), px * (12 + 8 + 12 + 12 + 8)),
    ("interp_fwd  SpatialTransformerSamplerInterp, warp field", lambda: ops.interp_fwd(src, gridp_warp), px * (12 + 8 + 12)),
    ("interp_bwd  (gx = 0, ggrid), warp field", lambda: ops.interp_bwd(src, gridp_warp, gy), px * (12 + 8 + 12 + 12 + 8)),
    ("interp_fwd  uniformly random grid", lambda: ops.interp_fwd(src, gridp), px * (12 + 8 + 12)),
    ("interp_bwd  uniformly random grid", lambda: ops.interp_bwd(src, gridp, gy), px * (12 + 8 + 12 + 12 + 8)),
    ("pyramid  4 scales, 6 planes", lambda: ops.pyramid(full_src, 4), int(B * 6 * P * 4 * (1 + 0.328))),
    ("pyramid_hwc  4 scales, 6 planes (scale 0 included)", lambda: ops.pyramid_hwc(full_src, 4), int(B * 6 * P * 4 * (1 + 1.328))),
    ("pyramid_pair_hwc  tgt + 2 src, 4 scales, one launch", lambda: ops.pyramid_pair_hwc(tgt, full_src, 4), int(B * 9 * P * 4 * (1 + 1.328))),
    ("disp_act_fwd  4 scales", lambda: ops.disp_act_fwd(logits), int(B * P * 1.328 * 8)),
    ("disp_act_bwd  4 scales", lambda: ops.disp_act_bwd(dd, dd), int(B * P * 1.328 * 12)),
]
print("operator kernels at B=%d, %dx%d (one launch each; output allocation by torch included)" % (B, H, W))
for name, fn, nbytes in rows:
    us = timeit(fn)
    print("%-60s %8.1f us %s" % (name, us, ("%7.0f GB/s algorithmic" % (nbytes / us / 1e3)) if nbytes else ""))
