#!/usr/bin/env python3
"""Pins the Chainer-defined arithmetic of the hot path -- for a machine that HAS the reference's dependency.

    pip install 'chainer==4.0.0b1' numpy          # requirements.txt:1 of pfnet/sfm-learner-chainer (CPU is enough)
    python tests/golden/make_chainer_golden.py [--reference /path/to/sfm-learner-chainer] [--out tests/golden]

It cannot run in the build container of this repository (no Chainer, no network) and nothing in the test-suite needs it:
the fixtures it writes (`chainer_*.npz`, data only: inputs + outputs of Chainer's own functions) are picked up by
`tests/test_oracle_cpu.py::test_oracle_matches_chainer_fixtures` and
`tests/test_ops_gpu.py::test_kernels_match_chainer_fixtures` WHEN PRESENT and turn the "parity unpinned" rows of DESIGN.md 3 into
pinned ones; when absent those tests are skipped with that reason.

What it records, with the reference call site each one stands for:
  chainer_sampler_*.npz   F.spatial_transformer_sampler(x, grid) forward, and backward to (gx, ggrid)      models/transform.py:189
  chainer_resize_*.npz    F.resize_images(x, (h, w))                                                       models/base_model.py:71-72
  chainer_pool_*.npz      F.average_pooling_2d(x, 3, 1, 1) forward + backward                              models/base_model.py:130-135
  chainer_matmul_*.npz    F.batch_matmul / F.batch_inv on (N,3,3) / (N,4,4) operands                       models/transform.py:39,88,105,122
  chainer_ssim_*.npz      SFMLearner.compute_ssim(x, y) (needs --reference)                                models/base_model.py:126-142
  chainer_warp_*.npz      projective_inverse_warp forward + backward (needs --reference)                   models/transform.py:156-193
  chainer_loss_*.npz      SFMLearner.__call__ itself, forward + backward, with DispNet / PoseNet replaced by callables that return
                          fixed disparities / poses (needs --reference)                                     models/base_model.py:48-124
  chainer_cfg1_l1.npz     BASELINE.json configs[0] as a whole: the reference's __call__ on ONE 128x416 3-frame snippet, 1 scale,
                          L1 only (experiments/sfm_learner_v1.yml) -> total_loss and the reported pixel_loss, the gradients, and the
                          warped image of source 0 (curr_proj_img, :90-94).  This ONE file turns "parity" green for the fused loss:
                          the oracle, sfm_loss_fwd_bwd and its warped-image output (SfmLossDesc.warped) are all compared with it.

Inputs come from sfm-learner-chainer_amd/synth.py with fixed seeds, so the consuming tests can rebuild them bit for bit and only
the OUTPUTS need to travel; they are stored anyway, to make the fixtures self-contained.
"""
import argparse
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("SFM_REFERENCE", ""), help="checkout of pfnet/sfm-learner-chainer (optional)")
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()
    try:
        import chainer
        import chainer.functions as F
    except ImportError:
        sys.exit("chainer is not importable here: install chainer==4.0.0b1 (requirements.txt:1 of the reference) and run again")
    synth = importlib.import_module("sfm-learner-chainer_amd.synth")
    print("chainer", chainer.__version__)
    rng = np.random.RandomState(7)

    def save(name, **arrays):
        path = os.path.join(args.out, name + ".npz")
        np.savez_compressed(path, chainer_version=np.array(chainer.__version__), **arrays)
        print("wrote", path)

    # ---- F.spatial_transformer_sampler: in-range, the zero-pad ring, far outside, exact corners
    for name, (N, C, H, W, oH, oW) in {"small": (2, 3, 8, 13, 8, 13), "ragged": (1, 2, 5, 7, 11, 3), "kitti_s3": (2, 3, 16, 52, 16, 52)}.items():
        x = rng.uniform(-1, 1, size=(N, C, H, W)).astype(np.float32)
        grid = rng.uniform(-1.6, 1.6, size=(N, 2, oH, oW)).astype(np.float32)
        grid[:, :, 0, 0] = [-1.0, 1.0]
        gy = rng.normal(size=(N, C, oH, oW)).astype(np.float32)
        xv, gv = chainer.Variable(x), chainer.Variable(grid)
        y = F.spatial_transformer_sampler(xv, gv)
        y.grad = gy
        y.backward()
        save("chainer_sampler_" + name, x=x, grid=grid, gy=gy, y=y.data, gx=xv.grad, ggrid=gv.grad)

    # ---- F.resize_images (the pyramid)
    for name, (N, C, H, W) in {"kitti": (2, 3, 128, 416), "odd": (1, 2, 37, 70)}.items():
        x = rng.uniform(-1, 1, size=(N, C, H, W)).astype(np.float32)
        outs = {}
        for s in (1, 2, 3):
            outs["y%d" % s] = F.resize_images(x, (H >> s, W >> s)).data
        save("chainer_resize_" + name, x=x, **outs)

    # ---- F.average_pooling_2d(x, 3, 1, 1)
    x = rng.uniform(-1, 1, size=(2, 3, 9, 14)).astype(np.float32)
    gy = rng.normal(size=x.shape).astype(np.float32)
    xv = chainer.Variable(x)
    y = F.average_pooling_2d(xv, 3, 1, 1)
    y.grad = gy
    y.backward()
    save("chainer_pool_3x3", x=x, gy=gy, y=y.data, gx=xv.grad)

    # ---- F.batch_matmul / F.batch_inv
    a = rng.normal(size=(5, 4, 4)).astype(np.float32)
    b = rng.normal(size=(5, 4, 7)).astype(np.float32)
    K = np.tile(np.array([[241.7, 0.3, 204.2], [0, 246.3, 59.0], [0, 0, 1]], np.float32), (5, 1, 1)) * rng.uniform(0.9, 1.1, size=(5, 1, 1)).astype(np.float32)
    save("chainer_matmul_inv", a=a, b=b, ab=F.batch_matmul(a, b).data, K=K, Kinv=F.batch_inv(K).data)

    if not args.reference:
        print("no --reference: the fixtures that call the reference's own functions are skipped")
        return
    sys.path.insert(0, args.reference)
    try:
        transform = importlib.import_module("models.transform")
    except Exception as e:   # e.g. cv2 / chainercv missing for base_model; transform only needs chainer
        sys.exit("could not import the reference's models.transform: %r" % (e,))

    # ---- projective_inverse_warp forward + backward (models/transform.py:156-193)
    for name, (B, H, W) in {"s3": (2, 16, 52), "s0": (1, 128, 416)}.items():
        d = synth.make_inputs(B=B, H=H, W=W, n_src=2, n_scales=1, seed=4)
        imgs = d["src_pyr"][0][:, :3].copy()
        depth = (1.0 / d["disps"][0]).reshape(B, 1, H * W).astype(np.float32)
        depthes = chainer.Variable(np.broadcast_to(depth, (B, 3, H * W)).copy())
        poses = chainer.Variable(d["poses"][0].copy())
        Kc = d["intrinsics"][:, 0].copy()
        y = transform.projective_inverse_warp(imgs, depthes, poses, Kc)
        g = rng.normal(size=y.shape).astype(np.float32)
        y.grad = g
        y.backward()
        save("chainer_warp_" + name, imgs=imgs, depthes=depthes.data, poses=poses.data, K=Kc, g=g, warped=y.data, d_depthes=depthes.grad, d_poses=poses.grad)

    # ---- compute_ssim and the loss loop need models.base_model (imports cv2 at module level: provide it or skip)
    try:
        base_model = importlib.import_module("models.base_model")
    except Exception as e:
        print("models.base_model not importable (%r): chainer_ssim / chainer_loss fixtures skipped" % (e,))
        return
    net = base_model.SFMLearner.__new__(base_model.SFMLearner)          # no DispNet / PoseNet: the loss path has no parameters
    chainer.Chain.__init__(net)
    x = rng.uniform(-1, 1, size=(2, 3, 16, 52)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(2, 3, 16, 52)).astype(np.float32)
    save("chainer_ssim_s3", x=x, y=y, ssim=net.compute_ssim(chainer.Variable(x), y).data)
    # the whole loss loop, by calling the reference's own SFMLearner.__call__ with the two networks replaced by callables that
    # hand back fixed disparities / poses (the timers need CuPy events even on the CPU: switched off)
    for name in ("create_timer", "print_timer"):
        if hasattr(base_model, name):
            setattr(base_model, name, lambda *a, **k: None)
    for cfg_name, cfg in {"l1_smooth": dict(smooth_reg=0.1, exp_reg=0.0, ssim_rate=0.0), "ssim_smooth": dict(smooth_reg=0.1, exp_reg=0.0, ssim_rate=0.15)}.items():
        d = synth.make_inputs(B=2, H=32, W=104, n_src=2, n_scales=4, seed=8)
        disps = [chainer.Variable(a.copy()) for a in d["disps"]]
        poses = [chainer.Variable(a.copy()) for a in d["poses"]]
        net.smooth_reg, net.exp_reg, net.ssim_rate, net.n_sources = cfg["smooth_reg"], cfg["exp_reg"], cfg["ssim_rate"], 2
        net.disp_net = lambda tgt, disps=disps: disps
        net.pose_net = lambda tgt, src, do_exp=False, poses=poses: (poses, None)
        total = net(d["tgt"], d["src"], d["intrinsics"], None)      # (chainer.report without a reporter in scope is a no-op)
        total.backward()
        save("chainer_loss_" + cfg_name, total=np.float32(total.data),
             **{"d_disp%d" % s: disps[s].grad for s in range(4)}, **{"d_pose%d" % i: poses[i].grad for i in range(2)})

    # ---- BASELINE.json configs[0]: the Chainer CPU reference's own case, as a whole (1 x (128x416) 3-frame snippet, 1 scale, L1 only)
    d = synth.make_inputs(B=1, H=128, W=416, n_src=2, n_scales=1, seed=1)
    disps = [chainer.Variable(d["disps"][0].copy())]
    poses = [chainer.Variable(a.copy()) for a in d["poses"]]
    net.smooth_reg, net.exp_reg, net.ssim_rate, net.n_sources = 0.0, 0.0, 0.0, 2
    net.disp_net = lambda tgt, disps=disps: disps
    net.pose_net = lambda tgt, src, do_exp=False, poses=poses: (poses, None)
    reporter, seen = chainer.Reporter(), {}
    reporter.add_observer("sfm", net)                                # the five chainer.report keys of :119-123 land in `seen`
    with reporter.scope(seen):
        total = net(d["tgt"], d["src"], d["intrinsics"], None)
    total.backward()
    depth = (1.0 / d["disps"][0]).reshape(1, 1, 128 * 416).astype(np.float32)
    warped0 = transform.projective_inverse_warp(d["src"][:, 0].copy(), chainer.Variable(np.broadcast_to(depth, (1, 3, 128 * 416)).copy()),
                                                chainer.Variable(d["poses"][0].copy()), d["intrinsics"][:, 0].copy())     # :90-94 for source 0
    save("chainer_cfg1_l1", total=np.float32(total.data), pixel=np.float32(chainer.cuda.to_cpu(getattr(seen.get("sfm/pixel_loss"), "data", seen.get("sfm/pixel_loss")))),
         warped0=warped0.data, d_disp0=disps[0].grad, d_pose0=poses[0].grad, d_pose1=poses[1].grad)


if __name__ == "__main__":
    main()
