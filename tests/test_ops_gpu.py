"""GPU parity of the operator-level entry points of the C ABI against (a) the golden vectors
produced by the reference's own code and (b) the CPU oracle."""
import glob
import os

import numpy as np
import pytest

from oracle import sfm_oracle as O
from oracle.parity import rel_l2
from util import assert_close_masked, parity_note, to_dev, to_np

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "interp_sampler_*.npz"))))
def test_interp_sampler_matches_reference_golden_bit_exact(ops, dev, path):
    """SpatialTransformerSamplerInterp._forward/_backward
    (models/spational_transformer_sampler_interp.py:32-149) -- outputs of the reference's code."""
    z = np.load(path)
    x, grid, gy = (to_dev(z[k], dev) for k in ("x", "grid", "gy"))
    y = to_np(ops.interp_fwd(x, grid))
    gx, ggrid = ops.interp_bwd(x, grid, gy)
    np.testing.assert_array_equal(y, z["y"])
    np.testing.assert_array_equal(to_np(ggrid), z["ggrid"])
    np.testing.assert_array_equal(to_np(gx), z["gx"])          # == 0 (:148)


def test_pose_proj_matches_oracle_and_odom_util_golden(ops, dev):
    z = np.load(os.path.join(GOLD, "euler_odom_util.npz"))
    r = z["r_xyz"].astype(np.float32)
    N = r.shape[0]
    rng = np.random.RandomState(0)
    pose = np.concatenate([r, rng.normal(0, 0.1, size=(N, 3)).astype(np.float32)], axis=1)
    K = np.tile(np.eye(3, dtype=np.float32), (N, 1, 1))
    proj = to_np(ops.pose_proj_fwd(to_dev(pose, dev), to_dev(K, dev)))
    # with K = I the upper-left 3x3 is euler2mat: compare with kitti_eval/odom_util.py:167-200
    np.testing.assert_allclose(proj[:, :3, :3], z["R"], atol=5e-7)
    np.testing.assert_allclose(proj[:, :3, 3], pose[:, 3:], atol=0)
    np.testing.assert_array_equal(proj[:, 3], np.tile(np.array([0, 0, 0, 1], np.float32), (N, 1)))
    K = np.tile(np.array([[241.7, 0, 204.2], [0, 246.3, 59.0], [0, 0, 1]], np.float32), (N, 1, 1))
    K[:, 0, 1] = rng.normal(0, 0.5, N)       # general (skewed) K
    proj = to_np(ops.pose_proj_fwd(to_dev(pose, dev), to_dev(K, dev)))
    want = O.proj_tgt_to_src(pose, K)
    np.testing.assert_allclose(proj, want, rtol=2e-6, atol=2e-5)
    g = rng.normal(size=(N, 4, 4)).astype(np.float32)
    got = to_np(ops.pose_proj_bwd(to_dev(pose, dev), to_dev(K, dev), to_dev(g, dev)))
    want = O.proj_tgt_to_src_backward(pose.astype(np.float64), K.astype(np.float64), g.astype(np.float64), np.float64)
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4 * np.abs(want).max())


WARP_TOL = 1e-4      # north_star: warped pixels within 1e-4 of the reference (relative to the image range), on EVERY texture


@pytest.mark.parametrize("shape", [(2, 3, 16, 52), (1, 3, 37, 70), (2, 1, 9, 11), (1, 5, 128, 416), (2, 3, 128, 416)])
@pytest.mark.parametrize("texture", ["smooth", "noise"])
@pytest.mark.parametrize("depth_rows", [1, 3])
def test_projective_inverse_warp_fwd_bwd(ops, synth, dev, shape, texture, depth_rows):
    """projective_inverse_warp (models/transform.py:156-193).  The operator keeps the reference's evaluation order
    (Pm . (D . K^-1 . pix, 1), +1e-10, normalise, x2, the sampler's de-normalisation; no fused multiply-adds), so:
      * warped pixels agree with the oracle to 1e-4 of the image range on image-like AND white-noise sources, at every
        pixel -- no knife-edge exclusion;
      * the set of exactly-zero (out-of-view) pixels is the oracle's, pixel for pixel;
      * the backward is compared with the oracle's hand-derived one (element-wise outside the pixels whose sample sits on
        a cell boundary of the bilinear lattice, where dI^/du itself jumps; their share is printed and bounded)."""
    N, C, H, W = shape
    d = synth.make_inputs(B=N, H=H, W=W, n_src=2, n_scales=1, seed=4)
    rng = np.random.RandomState(1)
    if texture == "noise":
        imgs = rng.uniform(-1, 1, size=(N, C, H, W)).astype(np.float32)
    else:
        imgs = np.concatenate([d["tgt"], d["src"].reshape(N, -1, H, W)], axis=1)[:, :C].copy()
    imgs[imgs == 0] = 0.5
    depth = (1.0 / d["disps"][0]).reshape(N, H * W).astype(np.float32)
    if depth_rows == 3:      # three independent depth rows (the operator's general form, transform.py:105-107)
        depthes = (depth[:, None] * rng.uniform(0.98, 1.02, size=(N, 3, 1))).astype(np.float32)
        dev_depth = depthes
    else:                    # one row of the reference's broadcast (base_model.py:82-84)
        depthes = np.broadcast_to(depth[:, None], (N, 3, H * W))
        dev_depth = depth
    pose, K = d["poses"][0], d["intrinsics"][:, 0]
    want, aux = O.projective_inverse_warp(imgs, depthes, pose, K, return_aux=True)
    targs = [to_dev(a, dev) for a in (imgs, dev_depth, pose, K)]
    got = to_np(ops.warp_fwd(*targs))
    err = float(np.abs(got.astype(np.float64) - want).max())
    zero_g, zero_w = (got == 0).all(1), (want == 0).all(1)
    parity_note("warp_fwd %s %s rows=%d: max |err| %.2e (tol %.0e of range 1), zero-set mismatches %d of %d px, pixels within 2e-5 "
                "of the (-1,1) test %d" % (shape, texture, depth_rows, err, WARP_TOL, int((zero_g != zero_w).sum()), zero_g.size,
                                           int((aux["margin"] < 2e-5).sum())))
    assert_close_masked(got, want, WARP_TOL, None, what="warped")
    assert not (zero_g != zero_w).any(), "the exactly-zero pixels differ from the oracle's"
    # Backward.  The gradient of the bilinear sample jumps where the sample crosses a cell boundary of the lattice; the two evaluations
    # take the same side (the coordinates are the reference's) except within rounding of the boundary itself.  Those pixels (named by
    # the ORACLE's margins; share printed and bounded) get a ZERO upstream gradient, so that no output depends on them -- and every
    # output is then judged by the loss suite's criteria: d_depth and d_src element-wise at 2e-3 of the array's maximum EVERYWHERE,
    # d_pose element-wise at 2e-3 AND 1e-3 in relative L2 (tests/test_loss_gpu.py), no allowance.
    kcell = ((aux["cell_margin"] < 2e-5) & ~zero_w)[:, None]
    parity_note("warp_bwd %s %s rows=%d: upstream gradient zeroed on the cell-boundary share %.4f%%" % (shape, texture, depth_rows, 100 * kcell.mean()))
    assert kcell.mean() < 1e-3
    g = (rng.normal(size=(N, C, H, W)) * ~kcell).astype(np.float32)
    w_dep, w_pose, w_src = O.projective_inverse_warp_backward(imgs, depthes, pose, K, g, want_gimgs=True)
    d_depth, d_pose, d_src = ops.warp_bwd(*targs, to_dev(g, dev), want_d_src=True)
    got_dep = to_np(d_depth).reshape(N, depth_rows, H, W)
    want_dep = (w_dep.sum(1, keepdims=True) if depth_rows == 1 else w_dep).reshape(N, depth_rows, H, W)
    assert_close_masked(got_dep, want_dep, 2e-3, None, what="d_depth")
    assert rel_l2(got_dep, want_dep) <= 1e-4, rel_l2(got_dep, want_dep)
    # (d_pose sums H*W signed terms driven by a white-noise upstream gradient: it cancels to ~sqrt(HW) of one term, and the fp32
    #  oracle's own summation order shows at that level: the fp64 oracle is the yardstick when the fp32 one misses)
    pose_err, pose_l2 = np.abs(to_np(d_pose) - w_pose).max() / np.abs(w_pose).max(), rel_l2(to_np(d_pose), w_pose)
    if pose_err > 2e-3 or pose_l2 > 1e-3:
        w64 = O.projective_inverse_warp_backward(imgs.astype(np.float64), np.asarray(depthes, np.float64), pose.astype(np.float64),
                                                 K.astype(np.float64), g.astype(np.float64), dtype=np.float64)[1]
        own_err, own_l2 = np.abs(w_pose - w64).max() / np.abs(w64).max(), rel_l2(w_pose, w64)
        pose_err, pose_l2 = np.abs(to_np(d_pose) - w64).max() / np.abs(w64).max(), rel_l2(to_np(d_pose), w64)
        parity_note("warp_bwd %s %s rows=%d: d_pose judged against the fp64 oracle (the fp32 oracle's own error: %.2e element-wise, %.2e L2)" % (
            shape, texture, depth_rows, own_err, own_l2))
        assert pose_err <= max(2e-3, 3 * own_err) and pose_l2 <= max(1e-3, 3 * own_l2), (pose_err, pose_l2, own_err, own_l2)
    parity_note("warp_bwd %s %s rows=%d: d_pose max element-wise %.2e (tol 2e-3), relative L2 %.2e (tol 1e-3)" % (shape, texture, depth_rows, pose_err, pose_l2))
    assert_close_masked(to_np(d_src), w_src, 2e-3, None, what="d_src")


@pytest.mark.parametrize("shape", [(2, 3, 8, 13, 8, 13), (1, 2, 5, 7, 11, 3), (2, 3, 16, 52, 16, 52), (1, 6, 9, 70, 35, 67), (1, 4, 40, 9, 33, 130)])
def test_normalized_sampler_fwd_bwd(ops, dev, shape):
    """F.spatial_transformer_sampler as called at models/transform.py:189, arbitrary grids
    including the zero-pad ring and far outside."""
    N, C, H, W, oH, oW = shape
    rng = np.random.RandomState(2)
    x = rng.uniform(-1, 1, size=(N, C, H, W)).astype(np.float32)
    grid = rng.uniform(-1.6, 1.6, size=(N, 2, oH, oW)).astype(np.float32)
    grid[:, :, 0, 0] = [-1.0, 1.0]
    gy = rng.normal(size=(N, C, oH, oW)).astype(np.float32)
    want = O.spatial_transformer_sampler(x, grid)
    got = to_np(ops.sampler_fwd(to_dev(x, dev), to_dev(grid, dev)))
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    wgx, wgg = O.spatial_transformer_sampler_backward(x, grid, gy)
    gx, gg = ops.sampler_bwd(to_dev(x, dev), to_dev(grid, dev), to_dev(gy, dev))
    np.testing.assert_allclose(to_np(gg), wgg, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(to_np(gx), wgx, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("shape", [(2, 3, 16, 52), (1, 2, 37, 70), (3, 3, 128, 416), (1, 6, 37, 70), (2, 4, 67, 130)])
def test_normalized_sampler_bwd_on_warp_fields(ops, dev, shape):
    """The grids this sampler sees in the path (models/transform.py:189): the identity lattice plus a smooth displacement.
    Here most lanes share a tap column with their neighbour and sampler_bwd merges the two contributions before the atomic
    add: row ends, image borders (the zero-pad ring) and the last, partial wavefront of a plane are all in these cases."""
    N, C, H, W = shape
    rng = np.random.RandomState(6)
    x = rng.uniform(-1, 1, size=(N, C, H, W)).astype(np.float32)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    grid = np.empty((N, 2, H, W), np.float32)
    for n in range(N):
        fu = 3.0 * np.sin(yy / 7.0 + n) + 2.0 * np.cos(xx / 11.0) + 0.37 - 1.5 * n      # a few pixels, leaves the image at the borders
        fv = 2.0 * np.cos(yy / 5.0) - 1.5 * np.sin(xx / 9.0 + n) + 0.61
        grid[n, 0] = (xx + fu) / (W - 1) * 2 - 1
        grid[n, 1] = (yy + fv) / (H - 1) * 2 - 1
    gy = rng.normal(size=(N, C, H, W)).astype(np.float32)
    wgx, wgg = O.spatial_transformer_sampler_backward(x, grid, gy)
    gx, gg = ops.sampler_bwd(to_dev(x, dev), to_dev(grid, dev), to_dev(gy, dev))
    # (ggrid is the image gradient times (W-1)/2: a difference of cancelling products, scaled up)
    np.testing.assert_allclose(to_np(gg), wgg, rtol=1e-4, atol=1e-5 * np.abs(wgg).max())
    np.testing.assert_allclose(to_np(gx), wgx, rtol=1e-4, atol=2e-5)


def test_interp_equals_normalized_sampler_in_range(ops, dev):
    """SURVEY.md §8(c) pin (4): A8 == A8' on in-range coordinates after normalise -> pixel."""
    rng = np.random.RandomState(3)
    N, C, H, W = 2, 3, 12, 20
    x = rng.uniform(-1, 1, size=(N, C, H, W)).astype(np.float32)
    u = rng.uniform(0.01, W - 1.01, size=(N, H, W)).astype(np.float32)
    v = rng.uniform(0.01, H - 1.01, size=(N, H, W)).astype(np.float32)
    pix = np.stack([u, v], axis=1)
    nrm = np.stack([u / ((W - 1) / 2.) - 1, v / ((H - 1) / 2.) - 1], axis=1).astype(np.float32)
    a = to_np(ops.interp_fwd(to_dev(x, dev), to_dev(pix, dev)))
    b = to_np(ops.sampler_fwd(to_dev(x, dev), to_dev(nrm, dev)))
    np.testing.assert_allclose(a, b, atol=2e-5)


@pytest.mark.parametrize("shape", [(2, 3, 128, 416), (1, 6, 37, 70)])
def test_resize_matches_oracle(ops, dev, shape):
    rng = np.random.RandomState(4)
    x = rng.uniform(-1, 1, size=shape).astype(np.float32)
    H, W = shape[2:]
    for s in (1, 2, 3):
        oh, ow = H // 2 ** s, W // 2 ** s
        want = O.resize_images(x, (oh, ow))
        got = to_np(ops.resize(to_dev(x, dev), (oh, ow)))
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("shape,n_scales", [((2, 3, 128, 416), 4), ((1, 6, 37, 70), 3), ((2, 3, 16, 24), 1)])
def test_pyramid_matches_oracle_resizes(ops, dev, shape, n_scales):
    """the whole loop head models/base_model.py:69-72 in one launch == per-scale F.resize_images"""
    rng = np.random.RandomState(5)
    x = rng.uniform(-1, 1, size=shape).astype(np.float32)
    H, W = shape[2:]
    outs = ops.pyramid(to_dev(x, dev), n_scales)
    assert len(outs) == n_scales
    np.testing.assert_array_equal(to_np(outs[0]), x)
    for s in range(1, n_scales):
        want = O.resize_images(x, (H >> s, W >> s))
        np.testing.assert_allclose(to_np(outs[s]), want, rtol=1e-5, atol=2e-6)
        np.testing.assert_array_equal(to_np(outs[s]), to_np(ops.resize(to_dev(x, dev), (H >> s, W >> s))))


@pytest.mark.parametrize("shape,n_scales", [((2, 3, 32, 48), 4), ((3, 6, 37, 70), 3), ((1, 12, 128, 416), 4), ((2, 3, 9, 11), 1),
                                            # the band kernel (W % 4 == 0): bands of 3 rows at W = 832, a ragged last band, the two-row image,
                                            # a tall narrow one, and a row too long for the band's LDS (per-pixel kernel)
                                            ((1, 6, 256, 832), 4), ((2, 3, 37, 72), 3), ((1, 3, 2, 8), 1), ((1, 3, 131, 20), 3),
                                            ((1, 3, 16, 4096), 2)])
def test_pyramid_hwc_is_the_planar_pyramid_interleaved(ops, dev, shape, n_scales):
    """sfm_pyramid_hwc_fwd: the values of sfm_pyramid_fwd (bit for bit), laid out (N,G,h,w,3), scale 0 included"""
    rng = np.random.RandomState(6)
    x = to_dev(rng.uniform(-1, 1, size=shape).astype(np.float32), dev)
    planar = ops.pyramid(x, n_scales)
    hwc = ops.pyramid_hwc(x, n_scales)
    assert len(hwc) == n_scales
    for s in range(n_scales):
        assert tuple(hwc[s].shape) == (shape[0], shape[1] // 3, shape[2] >> s, shape[3] >> s, 3)
        np.testing.assert_array_equal(to_np(hwc[s]), to_np(ops.to_hwc(planar[s])))
    with pytest.raises(TypeError):
        ops.pyramid_hwc(x[:, :2], n_scales)             # not RGB triples


@pytest.mark.parametrize("N,n_src,H,W,n_scales", [(2, 2, 32, 48, 4), (3, 1, 37, 70, 3), (1, 4, 128, 416, 4), (9, 3, 9, 11, 1)])
def test_pyramid_pair_is_the_two_pyramids_in_one_launch(ops, dev, N, n_src, H, W, n_scales):
    rng = np.random.RandomState(7)
    tgt = to_dev(rng.uniform(-1, 1, size=(N, 3, H, W)).astype(np.float32), dev)
    src = to_dev(rng.uniform(-1, 1, size=(N, 3 * n_src, H, W)).astype(np.float32), dev)
    yt, ys = ops.pyramid_pair_hwc(tgt, src, n_scales)
    for a, b in zip(yt, ops.pyramid_hwc(tgt, n_scales)):
        np.testing.assert_array_equal(to_np(a), to_np(b))
    for a, b in zip(ys, ops.pyramid_hwc(src, n_scales)):
        np.testing.assert_array_equal(to_np(a), to_np(b))
    with pytest.raises(TypeError):
        import torch
        ops.pyramid_pair_hwc(torch.cat([tgt, tgt], 1), src, n_scales)        # the target is one image


def test_band_and_per_pixel_pyramid_kernels_agree_bitwise(ops, dev):
    """The band kernel (every input pixel read once, through LDS) and the per-pixel kernel it replaces for aligned shapes
    (selected for ONE call by sfm_pyramid_variant(1)) compute every output pixel with the same statements."""
    rng = np.random.RandomState(8)
    tgt = to_dev(rng.uniform(-1, 1, size=(3, 3, 128, 416)).astype(np.float32), dev)
    src = to_dev(rng.uniform(-1, 1, size=(3, 6, 128, 416)).astype(np.float32), dev)
    band = [to_np(a).copy() for pyr in ops.pyramid_pair_hwc(tgt, src, 4) for a in pyr]
    per_pixel = [to_np(a).copy() for pyr in ops.pyramid_pair_hwc(tgt, src, 4, per_pixel=True) for a in pyr]
    again = [to_np(a).copy() for pyr in ops.pyramid_pair_hwc(tgt, src, 4) for a in pyr]     # the switch holds for one call only
    for a, b, c in zip(band, per_pixel, again):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(a, c)


def test_type_checks(ops, dev):
    """check_type_forward of the reference (spational_transformer_sampler_interp.py:11-24)."""
    import torch
    x = torch.zeros((2, 3, 4, 5), device=dev)
    with pytest.raises(TypeError):
        ops.interp_fwd(x, torch.zeros((2, 3, 4, 5), device=dev))        # grid.shape[1] != 2
    with pytest.raises(TypeError):
        ops.interp_fwd(x, torch.zeros((1, 2, 4, 5), device=dev))        # batch mismatch
    with pytest.raises(TypeError):
        ops.interp_fwd(x.double(), torch.zeros((2, 2, 4, 5), device=dev))
    with pytest.raises(TypeError):
        ops.interp_fwd(x.cpu(), torch.zeros((2, 2, 4, 5)))              # no CPU path
    with pytest.raises(TypeError):
        ops.warp_fwd(torch.zeros((1, 3, 2, 8), device=dev), torch.ones((1, 16), device=dev),
                     torch.zeros((1, 6), device=dev), torch.eye(3, device=dev)[None])   # H < 3
    y = ops.interp_fwd(torch.zeros((0, 3, 4, 5), device=dev), torch.zeros((0, 2, 4, 5), device=dev))
    assert tuple(y.shape) == (0, 3, 4, 5)                                # empty batch


def test_kernels_match_chainer_fixtures(ops, dev):
    """The operator kernels against fixtures produced by Chainer itself (tests/golden/make_chainer_golden.py, to be run where
    chainer==4.0.0b1 is installed).  Skipped while no such fixture is committed -- see DESIGN.md 3, "parity unpinned"."""
    files = sorted(glob.glob(os.path.join(GOLD, "chainer_*.npz")))
    if not files:
        pytest.skip("no chainer_*.npz fixtures (run tests/golden/make_chainer_golden.py where Chainer is installed)")
    for path in files:
        z, name = np.load(path), os.path.basename(path)
        if name.startswith("chainer_sampler_"):
            x, grid, gy = (to_dev(z[k], dev) for k in ("x", "grid", "gy"))
            np.testing.assert_allclose(to_np(ops.sampler_fwd(x, grid)), z["y"], rtol=1e-5, atol=1e-6, err_msg=name)
            gx, gg = ops.sampler_bwd(x, grid, gy)
            np.testing.assert_allclose(to_np(gg), z["ggrid"], rtol=1e-4, atol=1e-5, err_msg=name)
            np.testing.assert_allclose(to_np(gx), z["gx"], rtol=1e-4, atol=1e-5, err_msg=name)
        elif name.startswith("chainer_resize_"):
            H, W = z["x"].shape[2:]
            for s in (1, 2, 3):
                np.testing.assert_allclose(to_np(ops.resize(to_dev(z["x"], dev), (H >> s, W >> s))), z["y%d" % s], rtol=1e-5, atol=2e-6, err_msg=name)
        elif name.startswith("chainer_cfg1_"):
            # BASELINE.json configs[0] as a whole, from the reference's own SFMLearner.__call__: the FUSED launch (loss, gradients
            # and the warped image it computed the loss on) against Chainer's numbers
            import importlib
            synth = importlib.import_module("sfm-learner-chainer_amd.synth")
            d = synth.make_inputs(B=1, H=128, W=416, n_src=2, n_scales=1, seed=1)
            fl = ops.FusedLoss().bind([to_dev(a, dev) for a in d["tgt_pyr"]], [to_dev(a, dev) for a in d["src_pyr"]], to_dev(d["intrinsics"], dev),
                                      [to_dev(a, dev) for a in d["disps"]], [to_dev(a, dev) for a in d["poses"]], want_warped=True)
            loss5 = to_np(fl.forward_backward())
            assert abs(loss5[0] - float(z["total"])) <= 1e-4 * abs(float(z["total"])) and abs(loss5[1] - float(z["pixel"])) <= 1e-4 * abs(float(z["pixel"])), name
            np.testing.assert_allclose(to_np(fl.warped[0])[:, 0], z["warped0"], rtol=0, atol=1e-4, err_msg=name)
            np.testing.assert_allclose(to_np(fl.d_disps[0]), z["d_disp0"], rtol=0, atol=2e-3 * np.abs(z["d_disp0"]).max(), err_msg=name)
            for i in range(2):
                np.testing.assert_allclose(to_np(fl.d_poses[i]), z["d_pose%d" % i], rtol=0, atol=2e-3 * np.abs(z["d_pose%d" % i]).max(), err_msg=name)
        elif name.startswith("chainer_warp_"):
            args = [to_dev(z[k], dev) for k in ("imgs", "depthes", "poses", "K")]
            np.testing.assert_allclose(to_np(ops.warp_fwd(*args)), z["warped"], rtol=0, atol=1e-4, err_msg=name)
            d_depth, d_pose, _ = ops.warp_bwd(*args, to_dev(z["g"], dev))
            np.testing.assert_allclose(to_np(d_depth), z["d_depthes"], rtol=0, atol=1e-3 * np.abs(z["d_depthes"]).max(), err_msg=name)
            np.testing.assert_allclose(to_np(d_pose), z["d_poses"], rtol=0, atol=2e-2 * np.abs(z["d_poses"]).max(), err_msg=name)
