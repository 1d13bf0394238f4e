"""The bookkeeping of the GPU parity tests, exercised on the CPU with the oracle standing in for the kernel: a tool that
accepts a d_pose array must accept exactly what it claims to."""
import numpy as np

import pytest
import torch

from oracle import sfm_oracle as O
from test_loss_gpu import CONFIGS, GRAD_TOL, _check_grads, _oracle, knife_widths, pose_explained_by_discontinuities, rel_l2


def _case(synth):
    """The sweep case that showed the effect on the GPU (SFM_SWEEP_N=400): B=2, 41x76, 4 sources, edge-aware L1; sample 0,
    source 0, pixel (21,44) samples 4e-6 px from a row boundary of the bilinear lattice."""
    cfg = CONFIGS["edge_aware_l1"]
    d = synth.make_inputs(B=2, H=41, W=76, n_src=4, n_scales=1, seed=1021983024 % 10000, with_masks=True)
    return d, cfg, _oracle(d, cfg)


def test_a_cell_flip_is_found_and_named(synth):
    d, cfg, ref = _case(synth)
    assert ref["cell_margin"][0][0, 0, 21, 44] < 1e-5
    other = dict(d, disps=[a.copy() for a in d["disps"]])
    other["disps"][0][0, 0, 21, 44] *= np.float32(1 - 2e-4)          # the same function, that pixel in the cell above
    got = _oracle(other, cfg)["d_poses"][0]
    want = ref["d_poses"][0]
    assert np.abs(got - want).max() > GRAD_TOL * np.abs(want).max()   # misses the flat criterion as is ...
    w2, named = pose_explained_by_discontinuities(d, cfg, ref, 0, got)
    assert named == ["sample 0 scale 0 pixel (21,44)"]
    assert np.abs(got - w2).max() <= 1e-5 * np.abs(want).max() and rel_l2(got, w2) < 1e-5   # ... and is that pixel's other branch


def test_an_error_that_is_no_flip_is_not_explained(synth):
    d, cfg, ref = _case(synth)
    want = ref["d_poses"][0]
    got = want * np.float32(1.004)                                     # a scale error
    w2, named = pose_explained_by_discontinuities(d, cfg, ref, 0, got)
    assert not named or np.abs(got - w2).max() > GRAD_TOL * np.abs(want).max()
    rng = np.random.RandomState(0)
    got = want + (6e-3 * np.abs(want).max() * rng.standard_normal(want.shape)).astype(np.float32)   # noise
    w2, named = pose_explained_by_discontinuities(d, cfg, ref, 0, got)
    assert not named or np.abs(got - w2).max() > GRAD_TOL * np.abs(want).max()


class _Stand:
    """What _check_grads reads of a bound FusedLoss, filled from arrays (the oracle standing in for the kernel)."""

    def __init__(self, d_disps, d_poses):
        self.d_disps = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)) for a in d_disps]
        self.d_poses = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)) for a in d_poses]


def _ladder_case(synth):
    cfg = CONFIGS["ssim_smooth"]
    d = synth.make_inputs(B=2, H=32, W=104, n_src=2, n_scales=2, seed=3)
    ref = _oracle(d, cfg)
    ref64 = lambda: O.sfm_loss(d["tgt_pyr"], d["src_pyr"], d["intrinsics"], d["disps"], d["poses"], None, backward=True,
                               keep_warped=True, dtype=np.float64, **cfg)
    explain = lambda i, got: pose_explained_by_discontinuities(d, cfg, ref, i, got)
    return d, cfg, ref, dict(ref64=ref64, explain=explain, **knife_widths(d, ref))


def test_two_correct_evaluations_pass_every_criterion(synth):
    """The fp32 oracle against itself, and the fp64 oracle's gradients rounded to fp32 against the fp32 oracle: both are
    correct evaluations of the reference's function, and the whole ladder of _check_grads must accept them."""
    d, cfg, ref, kw = _ladder_case(synth)
    _check_grads(_Stand(ref["d_disps"], ref["d_poses"]), ref, 2, what="oracle vs itself", **kw)
    r64 = kw["ref64"]()
    _check_grads(_Stand(r64["d_disps"], r64["d_poses"]), ref, 2, what="fp64 oracle vs fp32 oracle", **kw)


@pytest.mark.parametrize("which", ["d_disps", "d_poses"])
@pytest.mark.parametrize("error", ["scale_1pct", "sign", "one_sample_scale_1pct"])
def test_a_wrong_gradient_fails_through_every_rung(synth, which, error):
    """A 1 % scale error and a sign error on a gradient array must FAIL with every allowance of _check_grads available to it
    (flat criterion -> in-view allowance -> fp64 second opinion on elements and on the L2 norm -> explanation by named
    knife-edge pixels): none of the rungs may turn into a way round the comparison."""
    d, cfg, ref, kw = _ladder_case(synth)
    g = {"d_disps": [a.copy() for a in ref["d_disps"]], "d_poses": [a.copy() for a in ref["d_poses"]]}
    tgt = g[which][0]
    if error == "scale_1pct":
        tgt *= np.float32(1.01)
    elif error == "sign":
        tgt *= np.float32(-1.0)
    else:
        tgt[0] *= np.float32(1.01)         # one sample of the batch only
    with pytest.raises(AssertionError):
        _check_grads(_Stand(g["d_disps"], g["d_poses"]), ref, 2, what="%s with a %s error" % (which, error), **kw)
