"""The bookkeeping of the GPU parity tests, exercised on the CPU with the oracle standing in for the kernel: a tool that
accepts a d_pose array must accept exactly what it claims to."""
import numpy as np

from test_loss_gpu import CONFIGS, GRAD_TOL, _oracle, pose_explained_by_discontinuities, rel_l2


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
