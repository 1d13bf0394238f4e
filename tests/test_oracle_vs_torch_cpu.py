"""The oracle against an INDEPENDENT implementation with an independent backward.

`oracle/sfm_oracle.py` restates the reference op for op in NumPy and carries a hand-derived backward.
Here the same loss -- models/transform.py:11-193 and models/base_model.py:57-185 of pfnet/sfm-learner-chainer -- is written a
second time with torch ops on the CPU in float64, and its gradients come from torch's AUTOGRAD, not from any formula of
this repository:

  F.spatial_transformer_sampler  -> torch.nn.functional.grid_sample(align_corners=True, padding_mode="zeros")
  F.average_pooling_2d(x, 3,1,1) -> avg_pool2d(3, 1, 1)  (count_include_pad: divide by 9 always)
  F.resize_images                -> interpolate(mode="bilinear", align_corners=True)
  F.batch_matmul / F.batch_inv   -> torch.matmul / torch.linalg.inv
  F.sigmoid_cross_entropy(x, 1)  -> softplus(-x)

This does not pin the oracle on Chainer 4.0.0b1 (that needs tests/golden/make_chainer_golden.py on a machine that has
it); it removes the risk that the hand-derived backward and the finite-difference spot checks share a blind spot, and it
holds the published semantics of the four Chainer ops the path uses against a second, widely used implementation of them.
Tolerance: 1e-10 relative (both sides are float64; measured agreement ~1e-14).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from oracle import sfm_oracle as O

TOL = 1e-10
DT = torch.float64


def euler2mat(r):
    """models/transform.py:11-40"""
    r = torch.clamp(r, -math.pi, math.pi)
    c, s = torch.cos(r), torch.sin(r)
    zeros, ones = torch.zeros_like(r[:, 0]), torch.ones_like(r[:, 0])
    zmat = torch.stack([c[:, 2], -s[:, 2], zeros, s[:, 2], c[:, 2], zeros, zeros, zeros, ones], dim=1).reshape(-1, 3, 3)
    ymat = torch.stack([c[:, 1], zeros, s[:, 1], zeros, ones, zeros, -s[:, 1], zeros, c[:, 1]], dim=1).reshape(-1, 3, 3)
    xmat = torch.stack([ones, zeros, zeros, zeros, c[:, 0], -s[:, 0], zeros, s[:, 0], c[:, 0]], dim=1).reshape(-1, 3, 3)
    return torch.matmul(torch.matmul(xmat, ymat), zmat)


def proj_tgt_to_src(vec, K):
    """models/transform.py:43-91"""
    N = vec.shape[0]
    filler = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=DT).reshape(1, 1, 4).repeat(N, 1, 1)
    T = torch.cat([torch.cat([euler2mat(vec[:, :3]), vec[:, 3:].reshape(N, 3, 1)], dim=2), filler], dim=1)
    K_ = torch.cat([torch.cat([K, torch.zeros((N, 3, 1), dtype=DT)], dim=2), filler], dim=1)
    return torch.matmul(K_, T)


def projective_inverse_warp(imgs, depthes, poses, K):
    """models/transform.py:94-193; depthes (N,3,H*W)"""
    N, _, H, W = imgs.shape
    proj = proj_tgt_to_src(poses, K)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=DT), torch.arange(W, dtype=DT), indexing="ij")
    pix = torch.stack([xs, ys, torch.ones_like(xs)], dim=0).reshape(1, 3, H * W).expand(N, 3, H * W)
    cam = depthes * torch.matmul(torch.linalg.inv(K), pix)                                          # :105-107
    cam = torch.cat([cam, torch.ones((N, 1, H * W), dtype=DT)], dim=1)                              # :108
    q = torch.matmul(proj, cam)                                                                     # :122
    z = q[:, 2:3] + 1e-10                                                                           # :123
    px = (q[:, 0:1] / z) / ((W - 1) / 2.) - 1                                                       # :124
    py = (q[:, 1:2] / z) / ((H - 1) / 2.) - 1                                                       # :125
    p = torch.cat([px, py], dim=1)
    inside = (p.detach() > -1) & (p.detach() < 1)                                                   # :128-131
    p = p * torch.where(inside, torch.ones_like(p), torch.full_like(p, 2.0))
    grid = p.reshape(N, 2, H, W).permute(0, 2, 3, 1)                                                # (N,H,W,2): x, y
    return TF.grid_sample(imgs, grid, mode="bilinear", padding_mode="zeros", align_corners=True)    # :189


def compute_ssim(x, y):
    """models/base_model.py:126-142 (target-side statistics detached: `.data`)"""
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    pool = lambda t: TF.avg_pool2d(t, 3, 1, 1)
    mu_x, mu_y = pool(x), pool(y).detach()
    sigma_x = pool(x ** 2) - mu_x ** 2
    sigma_y = pool(y ** 2).detach() - mu_y ** 2
    sigma_xy = pool(x * y) - mu_x * mu_y
    n = (2 * mu_x * mu_y + c1) * (2 * sigma_xy + c2)
    d = (mu_x ** 2 + mu_y ** 2 + c1) * (sigma_x + sigma_y + c2)
    return torch.clamp((1 - n / d) / 2, 0., 1.)


def gradient(t):
    return t[:, :, :, 1:] - t[:, :, :, :-1], t[:, :, 1:] - t[:, :, :-1]      # D_dx, D_dy


def compute_smooth_loss(d):
    """models/base_model.py:169-185"""
    dx, dy = gradient(d)
    dx2, dxdy = gradient(dx)
    dydx, dy2 = gradient(dy)
    return dx2.abs().mean() + dxdy.abs().mean() + dydx.abs().mean() + dy2.abs().mean()


def compute_disp_smooth(img, d):
    """models/base_model.py:144-155"""
    i_dx, i_dy = gradient(img)
    i_dx, i_dy = i_dx.mean(dim=1, keepdim=True), i_dy.mean(dim=1, keepdim=True)
    d_dx, d_dy = gradient(d)
    return (d_dx.abs() * torch.exp(-i_dx.abs())).mean() + (d_dy.abs() * torch.exp(-i_dy.abs())).mean()


def torch_loss(tgt_pyr, src_pyr, intrinsics, disps, poses, masks, smooth_reg=0.0, exp_reg=0.0, ssim_rate=0.0,
               smooth_mode="second_order"):
    """SFMLearner.__call__ from the pyramid onwards, models/base_model.py:57-124"""
    B = tgt_pyr[0].shape[0]
    n_src = len(poses)
    smooth_loss = exp_loss = pixel_loss = ssim_loss = torch.zeros((), dtype=DT)
    for ns in range(len(disps)):
        tgt, src = tgt_pyr[ns], src_pyr[ns]
        if smooth_reg:                                                                    # :75-80
            term = compute_smooth_loss(disps[ns]) if smooth_mode == "second_order" else compute_disp_smooth(tgt, disps[ns])
            smooth_loss = smooth_loss + (smooth_reg / (2 ** ns)) * term
        depth = (1. / disps[ns]).reshape(B, 1, -1).expand(B, 3, -1)                       # :60,:81-84
        K = intrinsics[:, ns]
        for i in range(n_src):
            proj = projective_inverse_warp(src[:, i * 3:(i + 1) * 3], depth, poses[i], K)  # :90-94
            err = (proj - tgt).abs()                                                      # :95
            mask = (proj.detach() == 0).all(dim=1, keepdim=True).expand_as(err)           # :96-97
            err = torch.where(mask, torch.zeros_like(err), err)                           # :98-100
            if exp_reg:                                                                   # :103-109
                logit = masks[ns][:, i:i + 1]
                exp_loss = exp_loss + exp_reg * TF.softplus(-logit).mean()               # :157-167
                pixel_loss = pixel_loss + (err * torch.sigmoid(logit).expand_as(err)).mean()
            else:
                pixel_loss = pixel_loss + err.mean()                                      # :111
                if ssim_rate:                                                             # :112-115
                    ssim_loss = ssim_loss + (compute_ssim(proj, tgt) * (1 - mask.to(DT))).mean()
    total = (1 - ssim_rate) * pixel_loss + ssim_rate * ssim_loss + smooth_loss + exp_loss  # :117-118
    return total, pixel_loss, smooth_loss, exp_loss, ssim_loss


CONFIGS = {
    "l1": dict(),
    "l1_smooth": dict(smooth_reg=0.1),
    "ssim_smooth": dict(smooth_reg=0.1, ssim_rate=0.15),
    "ssim_only": dict(ssim_rate=0.15),
    "edge_aware": dict(smooth_reg=0.1, ssim_rate=0.15, smooth_mode="edge_aware"),
    "edge_aware_l1": dict(smooth_reg=0.3, smooth_mode="edge_aware"),
    "explain": dict(smooth_reg=0.1, exp_reg=0.2),
    "explain_alpha": dict(smooth_reg=0.1, exp_reg=0.2, ssim_rate=0.15),
}


@pytest.mark.parametrize("name", sorted(CONFIGS))
@pytest.mark.parametrize("shape", [(2, 24, 40, 2, 2), (1, 17, 29, 3, 3)])
def test_oracle_loss_and_gradients_match_torch_autograd(synth, name, shape):
    B, H, W, n_src, n_scales = shape
    cfg = CONFIGS[name]
    d = synth.make_inputs(B=B, H=H, W=W, n_src=n_src, n_scales=n_scales, seed=31, with_masks=True)
    # larger motion than the synthetic default, so that a good share of the pixels leaves the view (zero fill, x2 rule, mask)
    rng = np.random.RandomState(5)
    d["poses"] = [p + rng.normal(0, 0.03, p.shape).astype(np.float32) * np.array([1, 1, 1, 4, 4, 4], np.float32) for p in d["poses"]]
    ref = O.sfm_loss(d["tgt_pyr"], d["src_pyr"], d["intrinsics"], d["disps"], d["poses"], d["masks"], backward=True,
                     dtype=np.float64, keep_warped=True, **cfg)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64))
    disps = [t(a).requires_grad_(True) for a in d["disps"]]
    poses = [t(a).requires_grad_(True) for a in d["poses"]]
    masks = [t(a).requires_grad_(True) for a in d["masks"]]
    out = torch_loss([t(a) for a in d["tgt_pyr"]], [t(a) for a in d["src_pyr"]], t(d["intrinsics"]), disps, poses, masks, **cfg)
    out[0].backward()
    for got, key in zip(out, ("total_loss", "pixel_loss", "smooth_loss", "exp_loss", "ssim_loss")):
        assert abs(float(got.detach()) - ref[key]) <= TOL * max(abs(ref[key]), 1e-12), (key, float(got.detach()), ref[key])
    # some pixels really are out of view / masked in this case: the x2 rule and the zero fill are exercised
    out_of_view = [float((w == 0).all(axis=2).mean()) for w in ref["warped"]]
    assert max(out_of_view) > 0.01, out_of_view

    def close(a, b, what):
        a, b = a.numpy() if a is not None else np.zeros_like(b), np.asarray(b, np.float64)
        assert np.abs(a - b).max() <= TOL * max(np.abs(b).max(), 1e-30), (what, np.abs(a - b).max(), np.abs(b).max())

    for s in range(n_scales):
        close(disps[s].grad, ref["d_disps"][s], "d_disp[%d]" % s)
        if cfg.get("exp_reg"):
            close(masks[s].grad, ref["d_masks"][s], "d_mask[%d]" % s)
    for i in range(n_src):
        close(poses[i].grad, ref["d_poses"][i], "d_pose[%d]" % i)


def test_oracle_sampler_matches_grid_sample():
    """F.spatial_transformer_sampler as restated by the oracle against grid_sample on a random grid that
    reaches beyond the image (the zero-padded ring included), forward and both gradients."""
    rng = np.random.RandomState(3)
    x = rng.uniform(-1, 1, (2, 3, 9, 13))
    grid = rng.uniform(-1.3, 1.3, (2, 2, 7, 11))
    gy = rng.normal(0, 1, (2, 3, 7, 11))
    y = O.spatial_transformer_sampler(x, grid, dtype=np.float64)
    gx, ggrid = O.spatial_transformer_sampler_backward(x, grid, gy, dtype=np.float64)
    xt, gt = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(grid).requires_grad_(True)
    yt = TF.grid_sample(xt, gt.permute(0, 2, 3, 1), mode="bilinear", padding_mode="zeros", align_corners=True)
    yt.backward(torch.from_numpy(gy))
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(gx, xt.grad.numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(ggrid, gt.grad.numpy(), rtol=0, atol=1e-11)


def test_oracle_resize_and_pooling_match_torch():
    """F.resize_images (align-corners bilinear) and F.average_pooling_2d(3,1,1) (divide by 9, zero padding) as the oracle restates them."""
    rng = np.random.RandomState(4)
    x = rng.uniform(-1, 1, (2, 3, 16, 28))
    for oh, ow in ((8, 14), (4, 7), (16, 28), (5, 9)):
        want = TF.interpolate(torch.from_numpy(x), size=(oh, ow), mode="bilinear", align_corners=True).numpy()
        np.testing.assert_allclose(O.resize_images(x, (oh, ow), dtype=np.float64), want, rtol=0, atol=1e-12)
    np.testing.assert_allclose(O.average_pooling_3x3(x), TF.avg_pool2d(torch.from_numpy(x), 3, 1, 1).numpy(), rtol=0, atol=1e-13)
