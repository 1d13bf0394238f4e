"""Edge cases of the fused loss on the GPU, against the oracle: strip / chunk / XCD-mapping
boundaries, extreme batch and source counts, degenerate geometry."""
import os

import numpy as np
import pytest

from oracle import sfm_oracle as O
from test_loss_gpu import CONFIGS, KEYS, _bind, _check_grads, _check_losses, _oracle
from util import to_np

pytestmark = pytest.mark.gpu
CFG = CONFIGS["ssim_smooth"]


@pytest.mark.parametrize("B,H,W,n_src,n_scales", [
    (1, 3, 3, 1, 1),        # smallest legal image
    (1, 5, 60, 2, 1),       # exactly one gradient strip (64 lanes - 2x2 halo)
    (1, 5, 61, 2, 1),       # one column into the second strip
    (1, 16, 120, 1, 1),     # two full strips
    (9, 16, 24, 2, 2),      # B >= 8: samples striped over the XCDs, one XCD owns two samples
    (8, 16, 24, 3, 1),      # exactly one sample per XCD
    (3, 33, 40, 8, 1),      # the maximum number of sources
    (1, 48, 64, 1, 5),      # five scales (48x64 ... 3x4)
    (2, 70, 36, 2, 2),      # several row chunks, narrow image
])
@pytest.mark.parametrize("layout", ["planar", "hwc"])
def test_shapes_at_the_boundaries(ops, synth, dev, B, H, W, n_src, n_scales, layout):
    d = synth.make_inputs(B=B, H=H, W=W, n_src=n_src, n_scales=n_scales, seed=21)
    ref = _oracle(d, CFG)
    fl = _bind(ops, dev, d, CFG, layout=layout)
    _check_losses(fl.forward(), ref)
    _check_losses(fl.forward_backward(), ref)
    _check_grads(fl, ref, n_src)


def test_eight_scales_are_accepted(ops, synth, dev):
    d = synth.make_inputs(B=1, H=384, W=512, n_src=1, n_scales=8, seed=2)     # 384x512 ... 3x4
    cfg = CONFIGS["l1_smooth"]
    ref = _oracle(d, cfg)
    fl = _bind(ops, dev, d, cfg)
    _check_losses(fl.forward_backward(), ref)


def test_empty_batch(ops, synth, dev):
    import torch
    d = synth.make_inputs(B=1, H=16, W=24, n_src=2, n_scales=1, seed=2)
    z = lambda a: torch.zeros((0,) + a.shape[1:], dtype=torch.float32, device=dev)
    fl = ops.FusedLoss(**CFG).bind([z(a) for a in d["tgt_pyr"]], [z(a) for a in d["src_pyr"]], z(d["intrinsics"]),
                                   [z(a) for a in d["disps"]], [z(a) for a in d["poses"]], norm_B=4)
    assert to_np(fl.forward_backward()).tolist() == [0.0] * 5


def test_everything_out_of_view_is_masked_not_nan(ops, synth, dev):
    """a translation that moves every pixel out of the source: I^ == 0 everywhere -> masked
    (models/base_model.py:96-100), zero photometric loss and gradients, nothing non-finite"""
    d = synth.make_inputs(B=2, H=16, W=52, n_src=2, n_scales=2, seed=5)
    for p in d["poses"]:
        p[:, 3] = 50.0
    ref = _oracle(d, CFG)
    fl = _bind(ops, dev, d, CFG)
    loss = to_np(fl.forward_backward())
    assert ref["pixel_loss"] == 0.0 and loss[1] == 0.0 and loss[4] == 0.0
    assert abs(loss[2] - ref["smooth_loss"]) <= 1e-4 * ref["smooth_loss"]
    for g, w in zip(fl.d_disps, ref["d_disps"]):         # only the smoothness term is left
        np.testing.assert_allclose(to_np(g), w, rtol=0, atol=2e-5 * np.abs(w).max())
    for g in fl.d_poses:
        assert not to_np(g).any()


def test_rotation_angles_are_clipped_at_pi(ops, synth, dev):
    """F.clip(r, -pi, pi) (models/transform.py:23): angles beyond +-pi behave like +-pi and get no gradient"""
    d = synth.make_inputs(B=2, H=16, W=52, n_src=1, n_scales=1, seed=5)
    d["poses"][0][0, 2] = 4.0          # rz > pi for sample 0
    ref = _oracle(d, CFG)
    fl = _bind(ops, dev, d, CFG)
    _check_losses(fl.forward_backward(), ref)
    g = to_np(fl.d_poses[0])
    assert g[0, 2] == 0.0 and ref["d_poses"][0][0, 2] == 0.0


def test_extreme_disparities(ops, synth, dev):
    """DispNet's output range is (0.01, 10.01) (models/disp_net.py:7-8): depth from 0.1 to 100"""
    d = synth.make_inputs(B=2, H=16, W=52, n_src=2, n_scales=1, seed=7)
    rng = np.random.RandomState(0)
    d["disps"][0][:] = np.where(rng.rand(*d["disps"][0].shape) < 0.5, 0.0101, 10.0).astype(np.float32)
    cfg = CONFIGS["l1"]
    ref = _oracle(d, cfg)
    fl = _bind(ops, dev, d, cfg)
    _check_losses(fl.forward_backward(), ref)
    for x in fl.d_disps + fl.d_poses:
        assert np.isfinite(to_np(x)).all()


def test_high_resolution_config5_shape(ops, synth, dev):
    """BASELINE.json configs[4]: 256x832, 4 scales, 5-frame snippet (4 sources), at B=1.  d_pose sums 283k signed per-pixel
    terms whose magnitudes exceed the sum by orders of magnitude, so fp32 evaluations of the SAME formula differ at the 1e-3
    level (the fp32 oracle included): where the flat criterion is missed the fp64 oracle decides (_judged64)."""
    d = synth.make_inputs(B=1, H=256, W=832, n_src=4, n_scales=4, seed=1)
    ref = _oracle(d, CFG)
    fl = _bind(ops, dev, d, CFG, layout="hwc")
    _check_losses(fl.forward_backward(), ref)
    ref64 = lambda: O.sfm_loss(d["tgt_pyr"], d["src_pyr"], d["intrinsics"], d["disps"], d["poses"], backward=True, dtype=np.float64, **CFG)
    _check_grads(fl, ref, 4, what="cfg5 B=1", ref64=ref64)


def test_image_just_below_the_interleaved_layout_limit(ops, synth, dev):
    """SFM_LAYOUT_HWC takes images of fewer than 2^24 / 12 pixels (byte offsets formed in fp32, 32-bit lane offsets of the
    range-checked gathers): 896 x 1536 = 1.376 Mpixel is 16.5 of the 16.78 MB; one scale, one source, against the oracle.
    (The margins of the knife mask scale with the coordinates: near u = 1536 one fp32 ulp is 1.2e-4 px and both evaluations
    carry several, so a sample within 1e-3 px of a lattice line may sit in either cell -- 1e-4 at W = 416 -- and the sampled
    value moves by that times the image slope: |I^ - I| below 1.5e-4 may take either sign -- 3e-5 at W = 416.)"""
    d = synth.make_inputs(B=1, H=896, W=1536, n_src=1, n_scales=1, seed=3)
    ref = _oracle(d, CFG)
    fl = _bind(ops, dev, d, CFG, layout="hwc")
    _check_losses(fl.forward_backward(), ref)
    ref64 = lambda: O.sfm_loss(d["tgt_pyr"], d["src_pyr"], d["intrinsics"], d["disps"], d["poses"], backward=True, dtype=np.float64, **CFG)
    _check_grads(fl, ref, 1, what="896x1536", ref64=ref64, cell_thr=1e-3, abs_thr=1.5e-4)


def _random_cases(n, seed=2024):
    """Ragged shapes (odd sizes, partial strips and chunks), batch sizes on both sides of the XCD-striping threshold, every
    source count and loss mode.  The smallest scale keeps at least 12 x 24 pixels: below that a single knife-edge pixel is
    percents of every quantity compared and the comparison says nothing (the degenerate sizes are covered, with the plain
    criteria, by test_shapes_at_the_boundaries)."""
    rng = np.random.RandomState(seed)
    cases = []
    names = sorted(CONFIGS)
    for k in range(n):
        n_scales = int(rng.randint(1, 4))
        H = int(rng.randint(12 << (n_scales - 1), 97))
        W = int(rng.randint(24 << (n_scales - 1), 201))
        cases.append((int(rng.randint(1, 11)), H, W, int(rng.randint(1, 5)), n_scales, names[int(rng.randint(len(names)))], int(rng.randint(1 << 30))))
    return cases


# SFM_SWEEP_N widens the sweep for a soak run
@pytest.mark.parametrize("B,H,W,n_src,n_scales,cfg_name,seed", _random_cases(int(os.environ.get("SFM_SWEEP_N", "16"))))
def test_random_shapes_and_modes(ops, synth, dev, B, H, W, n_src, n_scales, cfg_name, seed):
    """A seeded sweep; every case is judged by the SAME criteria as the fixed-shape tests (_check_losses, _check_grads: loss
    1e-4, gradients 2e-3 element-wise + relative L2, knife share capped).  A pixel on the strict `-1 < x < 1` test of
    transform.py:129 may be sampled in one fp32 evaluation and exactly 0 in the other; it can move a loss term by at most
    (6 + 3) / (3 B h w).  Inputs whose pixels within 8e-6 of that test (20 ulp of the normalised coordinate) could move the loss
    by more than 5e-4, or whose knife-edge pixels exceed the cap, are re-drawn with the next seed (up to 16 draws; a choice made
    from oracle quantities alone).  The loss comparison itself gets NO allowance unless the kernel demonstrably zeroed a pixel
    differently from the oracle: those pixels are COUNTED (count_in_view_mismatches: single-source L1-only launches against the
    oracle's), and only their reach -- never more than the oracle-side bound -- is added.  A d_pose array that misses its criteria
    is accepted only when the oracle, re-run with named knife-edge pixels pushed across the discontinuity they sit on, matches the
    kernel by the flat criteria (pose_explained_by_discontinuities).  All of it is reported."""
    from test_loss_gpu import count_in_view_mismatches, knife_cap, knife_mask, knife_widths
    from util import parity_note
    cfg = CONFIGS[cfg_name]
    # (SFM_SWEEP_D_SRC=1: the same sweep with the optional dL/d(src) bound -- a soak of the two-launch d_src over shapes, batch sizes,
    #  source counts, loss modes, layouts and, with SFM_SWEEP_PROJECTION, both projections)
    want_src = os.environ.get("SFM_SWEEP_D_SRC") == "1"
    for attempt in range(16):
        d = synth.make_inputs(B=B, H=H, W=W, n_src=n_src, n_scales=n_scales, seed=(seed + attempt) % 10000, with_masks=True)
        ref = _oracle(d, cfg, want_d_src=want_src)
        flips = sum(int((ref["margin"][s_] < 8e-6).sum()) for s_ in range(n_scales))
        flip_reach = sum(float((ref["margin"][s_] < 8e-6).sum()) * 3.0 / (B * ref["margin"][s_].shape[-2] * ref["margin"][s_].shape[-1])
                         for s_ in range(n_scales))
        over = False
        for s_ in range(n_scales):
            m = knife_mask(ref, s_)[0]            # the mask _check_grads will exclude, same thresholds and footprints
            over |= m.mean() > 0.8 * knife_cap(m.shape[-2] * m.shape[-1])
        if flip_reach <= 5e-4 and not over:
            break
    else:
        pytest.fail("no admissible input in 16 draws")
    parity_note("sweep case B=%d %dx%d n_src=%d scales=%d %s: input re-drawn %d times; %d pixels on the (-1,1) test, reach %.1e of the loss" % (
        B, H, W, n_src, n_scales, cfg_name, attempt, flips, flip_reach))
    layout = "hwc" if seed % 2 else "planar"                                   # both image layouts take part in the sweep
    # (SFM_SWEEP_PROJECTION=reference_order runs the same sweep with SfmLossDesc.projection = SFM_PROJECTION_REFERENCE_ORDER: a soak of
    #  that mode over shapes, batch sizes, source counts, loss modes and layouts)
    fl = _bind(ops, dev, d, cfg, layout=layout, projection=os.environ.get("SFM_SWEEP_PROJECTION", "fast"), want_d_src=want_src)
    plain = dict(d, masks=None)
    count_in_view_mismatches(ops, dev, plain, ref, layout, "sweep %s %dx%d" % (cfg_name, H, W))
    counted = sum(c * 3.0 / (B * ref["margin"][s_].shape[-2] * ref["margin"][s_].shape[-1])
                  for s_, c in enumerate(count_in_view_mismatches.per_scale))
    slack = min(flip_reach, counted)
    parity_note("sweep case %s %dx%d: loss allowance %.1e (%d pixels counted as zeroed differently; oracle-side bound %.1e)" % (
        cfg_name, H, W, slack, sum(count_in_view_mismatches.per_scale), flip_reach))
    _check_losses(fl.forward(), ref, slack=slack)
    _check_losses(fl.forward_backward(), ref, slack=slack)
    ref64 = lambda: O.sfm_loss(d["tgt_pyr"], d["src_pyr"], d["intrinsics"], d["disps"], d["poses"], d["masks"], backward=True,
                               want_d_src=want_src, dtype=np.float64, **cfg)
    from test_loss_gpu import pose_explained_by_discontinuities
    _check_grads(fl, ref, n_src, check_src=want_src, check_mask=bool(cfg.get("exp_reg")), what="sweep %s %dx%d" % (cfg_name, H, W), ref64=ref64,
                 explain=lambda i, got: pose_explained_by_discontinuities(d, cfg, ref, i, got),
                 # (d_src is compared element-wise without a second opinion: like the d_src tests, the soak takes the knife classes at
                 #  the width the fp32 uncertainty of each sample's position gives them -- tools/diag_sweep_dsrc.py: where the flat widths
                 #  leave an element off, the kernel agrees with the fp64 oracle to the last digit and the fp32 oracle does not)
                 **(knife_widths(d, ref) if want_src else {}))


@pytest.mark.parametrize("B,H,W,n_src,n_scales,cfg_name", [(2, 14, 30, 2, 1, "ssim_smooth"), (3, 16, 26, 3, 1, "edge_aware"), (1, 12, 40, 2, 1, "l1_smooth")])
def test_inadmissible_inputs_are_right_where_they_can_be(ops, synth, dev, B, H, W, n_src, n_scales, cfg_name):
    """The sweep above re-draws an input whose knife-edge share is inadmissible (round-4 verdict: "the suite never shows what the
    kernel does on an inadmissible input").  Here such inputs are SOUGHT -- tiny frames, where a handful of pixels on the strict
    in-view test is percents of everything -- and the kernel is held to what holds on any input: the loss within 1e-4 plus the
    ORACLE-side bound of what the pixels on the strict test can move it by; warped pixels zeroed differently only within 8e-6 of
    that test and equal elsewhere by the criterion of _check_warped; every gradient finite; d_disp element-wise 2e-3 outside the
    knife mask -- with NO cap on the mask's share, which is printed -- and d_pose within 5 % in relative L2 (its sums carry the
    flipped pixels whole)."""
    from test_loss_gpu import _check_warped, knife_cap, knife_mask, GRAD_TOL
    from oracle.parity import rel_l2
    from util import parity_note
    cfg = CONFIGS[cfg_name]
    found = None
    for seed in range(300, 700):
        d = synth.make_inputs(B=B, H=H, W=W, n_src=n_src, n_scales=n_scales, seed=seed)
        ref = O.sfm_loss(d["tgt_pyr"], d["src_pyr"], d["intrinsics"], d["disps"], d["poses"], None, backward=True, keep_warped=True, **cfg)
        flip_reach = sum(float((ref["margin"][s_] < 8e-6).sum()) * 3.0 / (B * ref["margin"][s_].shape[-2] * ref["margin"][s_].shape[-1])
                         for s_ in range(n_scales))
        over = any(knife_mask(ref, s_)[0].mean() > knife_cap(H * W) for s_ in range(n_scales))
        if flip_reach > 5e-4 or over:
            found = (seed, d, ref, flip_reach)
            break
    assert found is not None, "no inadmissible input among 400 seeds: the case no longer tests anything"
    seed, d, ref, flip_reach = found
    fl = _bind(ops, dev, d, cfg, layout="hwc", want_warped=True)
    got = to_np(fl.forward_backward())
    for k, name in enumerate(KEYS):
        want = ref[name]
        assert abs(got[k] - want) <= 1e-4 * max(abs(want), 1e-6) + flip_reach, (name, got[k], want, flip_reach)
    n_flip = _check_warped(fl, ref, "INADMISSIBLE %s B=%d %dx%d seed %d" % (cfg_name, B, H, W, seed), d)
    shares = []
    for s_ in range(n_scales):
        m = knife_mask(ref, s_)[0][:, None]
        shares.append(float(m.mean()))
        g, w = to_np(fl.d_disps[s_]), ref["d_disps"][s_]
        assert np.isfinite(g).all()
        err = np.abs(g - w) * ~m
        assert err.max() <= GRAD_TOL * np.abs(w).max(), (s_, err.max() / np.abs(w).max())
    l2 = max(rel_l2(to_np(g), w) for g, w in zip(fl.d_poses, ref["d_poses"]))
    assert all(np.isfinite(to_np(g)).all() for g in fl.d_poses) and l2 <= 0.05, l2
    parity_note("INADMISSIBLE input %s B=%d %dx%d seed %d: pixels on the strict test can move the loss by %.1e (admissible: 5e-4), knife share %s; "
                "kernel: loss within 1e-4 + that bound, %d pixels zeroed differently, d_disp 2e-3 outside the (uncapped) mask, d_pose relative L2 %.1e" % (
                    cfg_name, B, H, W, seed, flip_reach, ", ".join("%.2f %%" % (100 * v) for v in shares), n_flip, l2))


@pytest.mark.parametrize("rows", [4, 28])
def test_forced_chunk_heights(rows):
    """The planner's extremes, forced through SFM_CHUNK_ROWS in a child process (the override is read once per process):
    4-row chunks (more halo rows than rows) and 28-row chunks (32 steps per pass: every bit of the step masks)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, SFM_CHUNK_ROWS=str(rows))
    r = subprocess.run([sys.executable, os.path.join(here, "forced_chunks_probe.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and ("OK rows=%d" % rows) in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("n_src,cfg_name,layout", [(5, "ssim_smooth", "hwc"), (7, "l1_smooth", "planar"), (8, "edge_aware", "hwc"), (8, "explain", "planar")])
def test_up_to_eight_sources(ops, synth, dev, n_src, cfg_name, layout):
    """SFM_MAX_SRC = 8 sources: since round 4 every wave builds the geometry of ALL its sources at once, eight lanes per source
    (build_wave_geom: lanes 8g .. 8g+7 belong to source g) -- at eight sources the sixty-four lanes are exactly used up, and no
    BASELINE config or sweep case goes beyond four.  Loss, warped pixels of every source and every gradient against the oracle."""
    from test_loss_gpu import _check_warped, knife_widths
    cfg = CONFIGS[cfg_name]
    d = synth.make_inputs(B=3, H=24, W=70, n_src=n_src, n_scales=2, seed=40 + n_src, with_masks=True)
    ref = _oracle(d, cfg)
    fl = _bind(ops, dev, d, cfg, layout=layout, want_warped=True)
    _check_losses(fl.forward_backward(), ref)
    what = "%d sources %s %s" % (n_src, cfg_name, layout)
    _check_warped(fl, ref, what, d)
    _check_grads(fl, ref, n_src, check_mask=bool(cfg.get("exp_reg")), what=what, **knife_widths(d, ref))
    assert len(fl.d_poses) == n_src and all(np.abs(to_np(g)).max() > 0 for g in fl.d_poses)      # every source got its own pose gradient


def test_sweep_rarely_needs_the_last_rungs():
    """Runs after the sweep (definition order): the explanation by named knife-edge pixels is a last resort, not a way of life --
    if more than 2 % of the sweep's cases (at least one) needed it for a d_pose array, or more than 10 % the fp64 second opinion,
    something systematic is off and the sweep FAILS instead of only noting it (round-3 advisor finding).  Measured at 400 cases in
    round 4: 2 arrays explained (0.5 % of cases), 14 fp64 uses (3.5 %)."""
    import util
    rows = [r for r in util.PARITY_ROWS if str(r.get("case", "")).startswith("sweep ")]
    if not rows:
        pytest.skip("the sweep did not run in this session")
    cases = {r["case"] for r in rows}
    explained = {r["case"] for r in rows if "explained" in r["elementwise_rung"]}
    fp64 = {r["case"] for r in rows if "fp64" in r["elementwise_rung"] or "fp64" in r["l2_rung"]}
    assert len(explained) <= max(1, 0.02 * len(cases)), "%d of %d sweep cases needed the explanation rung: %s" % (len(explained), len(cases), sorted(explained))
    assert len(fp64) <= max(2, 0.10 * len(cases)), "%d of %d sweep cases needed the fp64 second opinion: %s" % (len(fp64), len(cases), sorted(fp64))
