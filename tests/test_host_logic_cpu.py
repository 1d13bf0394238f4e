"""Host-side logic that needs no GPU: the Chainer-surface (Variable / Function protocol /
type_check / argument / report / no_backprop_mode), the reference-named wrappers' argument
handling, batch sharding, the bench plumbing.  No kernel is launched."""
import importlib

import os

import numpy as np
import pytest
import torch

cs = importlib.import_module("sfm-learner-chainer_amd.chainer_surface")
fn = importlib.import_module("sfm-learner-chainer_amd.functions")
links = importlib.import_module("sfm-learner-chainer_amd.links")
dist_mod = importlib.import_module("sfm-learner-chainer_amd.dist")
ops = importlib.import_module("sfm-learner-chainer_amd.ops")


class _Scale(cs.Function):          # a toy Function with a CPU forward, only to exercise the protocol
    def __init__(self, k):
        self.k = k

    def check_type_forward(self, in_types):
        cs.type_check.expect(in_types.size() == 1, in_types[0].dtype.char == 'f')

    def forward_cpu(self, inputs):
        return inputs[0] * self.k,

    def backward_cpu(self, inputs, grad_outputs):
        return grad_outputs[0] * self.k,


class _SumAll(cs.Function):
    def forward_cpu(self, inputs):
        return (inputs[0].sum() + inputs[1].sum()).reshape(()),

    def backward_cpu(self, inputs, grad_outputs):
        return grad_outputs[0] * torch.ones_like(inputs[0]), grad_outputs[0] * torch.ones_like(inputs[1])


def test_function_protocol_and_backward_over_a_dag():
    x = cs.Variable(torch.arange(4, dtype=torch.float32))
    a = _Scale(2.0)(x)
    b = _Scale(3.0)(x)
    loss = _SumAll()(a, b)
    assert loss.creator is not None and loss.rank == 2
    loss.backward()
    np.testing.assert_array_equal(x.grad.numpy(), np.full(4, 5.0, np.float32))   # d/dx (2x + 3x)
    x.cleargrad()
    assert x.grad is None


def test_non_scalar_backward_needs_a_seed_gradient():
    x = cs.Variable(torch.ones(3))
    y = _Scale(2.0)(x)
    with pytest.raises(RuntimeError):
        y.backward()
    y.grad = torch.full((3,), 0.5)
    y.backward()
    np.testing.assert_array_equal(x.grad.numpy(), np.ones(3, np.float32))


def test_no_backprop_mode_builds_no_graph():
    x = cs.Variable(torch.ones(3))
    with cs.no_backprop_mode():
        y = _Scale(2.0)(x)
    assert y.creator is None
    assert cs.config.enable_backprop is True


def test_type_check_and_raw_array_inputs():
    with pytest.raises(cs.InvalidType):
        _Scale(2.0)(cs.Variable(torch.ones(3, dtype=torch.float64)))
    y = _Scale(2.0)(torch.ones(3))            # raw arrays are accepted like Variables
    assert isinstance(y, cs.Variable) and y.creator is None
    with pytest.raises(TypeError):
        _Scale(2.0)(np.ones(3, np.float32))   # numpy arrays are not device arrays


def test_interp_wrapper_rejects_kwargs_like_the_reference():
    x = torch.zeros((1, 3, 4, 5))
    g = torch.zeros((1, 2, 4, 5))
    with pytest.raises(ValueError, match="use_cudnn"):
        fn.spatial_transformer_sampler_interp(x, g, use_cudnn=True)       # :153-157
    with pytest.raises(TypeError):
        fn.spatial_transformer_sampler_interp(x, g, bogus=1)              # :158


def test_cpu_arrays_fail_loudly_no_fallback():
    x = torch.zeros((1, 3, 4, 5))
    g = torch.zeros((1, 2, 4, 5))
    with pytest.raises(NotImplementedError, match="no CPU fallback"):
        fn.spatial_transformer_sampler_interp(x, g)
    with pytest.raises(NotImplementedError, match="no CPU fallback"):
        fn.projective_inverse_warp(torch.zeros((1, 3, 4, 5)), torch.ones((1, 3, 20)), torch.zeros((1, 6)), torch.eye(3)[None])
    with pytest.raises(TypeError, match="GPU-only"):
        ops.interp_fwd(x, g)


def test_sampler_type_check_conditions():
    """spational_transformer_sampler_interp.py:11-24, checked before any dispatch"""
    f = fn.SpatialTransformerSamplerInterp()
    with pytest.raises(cs.InvalidType):
        f(torch.zeros((2, 3, 4, 5)), torch.zeros((2, 3, 4, 5)))           # grid.shape[1] != 2
    with pytest.raises(cs.InvalidType):
        f(torch.zeros((2, 3, 4, 5)), torch.zeros((1, 2, 4, 5)))           # batch mismatch
    with pytest.raises(cs.InvalidType):
        f(torch.zeros((2, 3, 4, 5), dtype=torch.float64), torch.zeros((2, 2, 4, 5)))
    with pytest.raises(cs.InvalidType):
        f(torch.zeros((3, 4, 5)), torch.zeros((2, 2, 4, 5)))              # ndim


def test_loss_link_constructor_keys():
    link = links.SFMLearnerLoss({"smooth_reg": 0.1, "exp_reg": 0, "seq_len": 3})     # sfm_learner_v1.yml:13-16
    assert (link.smooth_reg, link.exp_reg, link.ssim_rate, link.n_sources) == (0.1, 0, 0.0, 2)
    link = links.SFMLearnerLoss({"smooth_reg": 0.1, "exp_reg": 0, "seq_len": 5, "ssim_rate": 0.15})
    assert (link.ssim_rate, link.n_sources) == (0.15, 4)
    with pytest.raises(KeyError):
        links.SFMLearnerLoss({"smooth_reg": 0.1, "seq_len": 3})
    assert links.parse_dict(None, "a", 3) == 3 and links.parse_dict({"a": 1}, "a", 3) == 1


def test_report_keeps_the_reference_keys():
    cs.report({"total_loss": 1.0}, None)
    cs.report({"pixel_loss": 2.0}, None)
    r = cs.get_report()
    assert r["total_loss"] == 1.0 and r["pixel_loss"] == 2.0


def test_shard_range_partitions_the_batch():
    for B in (1, 7, 32, 256):
        for w in (1, 2, 3, 8):
            spans = [dist_mod.shard_range(B, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_fused_loss_rejects_bad_configuration_before_touching_the_gpu():
    with pytest.raises(ValueError):
        ops.FusedLoss(smooth_mode="bogus")
    fl = ops.FusedLoss(ssim_rate=0.15)
    with pytest.raises(TypeError):
        fl.bind([torch.zeros((1, 3, 8, 8))], [torch.zeros((1, 6, 8, 8))], torch.zeros((1, 1, 3, 3)),
                [torch.ones((1, 1, 8, 8))], [torch.zeros((1, 6))] * 2)      # CPU tensors


def test_bench_workloads_follow_baseline_json():
    import json
    import os
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert "B=32" in base["configs"][2] and "128" in base["configs"][2]
    B, H, W, n_src, n_scales, cfg, _ = bench.WORKLOADS["cfg3"]
    assert (B, H, W, n_src, n_scales) == (32, 128, 416, 2, 4) and cfg["ssim_rate"] == 0.15 and cfg["smooth_reg"] == 0.1
    assert bench.BYTES_FWD + bench.BYTES_BWD == 60
    px = B * n_src * sum((H >> s) * (W >> s) for s in range(n_scales))
    assert px == 4526080
    # the headline workload is BASELINE configs[2] AS WRITTEN: L1 + SSIM + EDGE-AWARE smoothness (models/base_model.py:144-155)
    assert "edge-aware" in base["configs"][2]
    import argparse
    default = [a for a in open(os.path.join(root, "bench.py")) if '"--workload"' in a][0]
    assert 'default="cfg3_edge"' in default
    Be, He, We, ne, se, cfge, desc = bench.WORKLOADS["cfg3_edge"]
    assert (Be, He, We, ne, se) == (32, 128, 416, 2, 4) and cfge["smooth_mode"] == "edge_aware" and cfge["ssim_rate"] == 0.15
    assert bench.kernel_symbol(cfge, "hwc", "fused") == "void sfm::loss_kernel<true, true, true, false, 2, true, false>(sfm::LossArgs)"


def test_augmentation_parameters_follow_the_reference_rng_order():
    """datasets/kitti/kitti_raw_transformed.py:34,:50-51,:64: uniform(1,1.15,2), randint, randint, rand"""
    aug = importlib.import_module("sfm-learner-chainer_amd.augment")
    H, W = 128, 416
    p = aug.sample_params(np.random.RandomState(3), 3, H, W)
    rng = np.random.RandomState(3)
    for b in range(3):
        sc = rng.uniform(1, 1.15, 2)
        sh, sw = int(H * sc[1]), int(W * sc[0])
        oy = int(rng.randint(0, sh - H + 1))
        ox = int(rng.randint(0, sw - W + 1))
        flip = rng.rand() < 0.5
        assert tuple(p[b]) == (sc[0], sc[1], sh, sw, oy, ox, float(flip))
        assert 1.0 <= p[b, 0] < 1.15 and 0 <= oy <= sh - H and 0 <= ox <= sw - W
    K = np.tile(np.array([[241.7, 0, 204.2], [0, 246.3, 59.0], [0, 0, 1]], np.float32), (3, 1, 1))
    K2 = aug.augment_intrinsics(K, p, W)
    assert K2.shape == (3, 3, 3) and (K2[:, 2, 2] == 1).all() and (K2[:, 0, 0] >= K[:, 0, 0]).all()
    Kms = aug.get_multi_scale_intrinsics(K2, 4)
    np.testing.assert_allclose(Kms[:, 2, 0, 0], K2[:, 0, 0] / 4)


def test_augmentation_host_side_matches_the_reference_run_golden():
    """PINNED (round 6): augment.py's draws and intrinsics against tests/golden/intrinsics_aug.npz, produced by executing the
    reference's own datasets/kitti/kitti_raw_transformed.py:17-93 (tests/golden/make_golden.py): scaled size, crop offsets, flip
    decision, the intrinsics after augmentation and the multi-scale intrinsics, bit for bit."""
    aug = importlib.import_module("sfm-learner-chainer_amd.augment")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "intrinsics_aug.npz"))
    for k in range(int(z["n_cases"])):
        g = lambda name: z["c%d_%s" % (k, name)]
        H, W = [int(v) for v in g("hw")]
        p = aug.sample_params(np.random.RandomState(int(g("seed"))), 1, H, W)
        assert (int(p[0, 2]), int(p[0, 3])) == tuple(int(v) for v in g("scaled_hw"))
        assert (int(p[0, 4]), int(p[0, 5])) == tuple(int(v) for v in g("offset_yx")) and bool(p[0, 6]) == bool(g("flip"))
        K = aug.augment_intrinsics(g("K_in")[None], p, W)
        np.testing.assert_array_equal(K[0], g("K_out"))
        np.testing.assert_array_equal(aug.get_multi_scale_intrinsics(K, g("K_multi").shape[0])[0], g("K_multi"))


def test_rccl_binding_finds_the_library_torch_loaded():
    """rccl.py binds the librccl.so instance that torch has mapped (the one sharing torch's HIP runtime), declares the four entry
    points bench.py uses and can ask it for a unique id without a GPU; communicators and collectives need devices (GPU test:
    tests/test_bench_rehearsal_gpu.py::test_drivers_launcher_form_runs_on_real_rccl_at_one_rank)."""
    import ctypes as C
    import importlib
    rccl = importlib.import_module("sfm-learner-chainer_amd.rccl")
    path = rccl._loaded_librccl()
    # torch's own copy where torch ships one (the copy that shares its HIP runtime), else the system library torch links
    import torch
    own = os.path.join(os.path.dirname(os.path.abspath(torch.__file__)), "lib", "librccl.so")
    assert path and (os.path.dirname(os.path.abspath(path)) == os.path.dirname(own) if os.path.exists(own) else "librccl" in path), path
    L = rccl.lib()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclCommDestroy", "ncclGetErrorString"):
        assert hasattr(L, name), name
    uid = rccl._UniqueId()
    assert L.ncclGetUniqueId(C.byref(uid)) == 0 and C.sizeof(uid) == 128
    raw = rccl._id_bytes(uid)
    assert len(raw) == 128 and any(raw) and 0 in raw          # (an id holds NUL bytes: it must travel as 128 raw bytes, not as a C string)
    back = rccl._UniqueId()
    C.memmove(C.byref(back), raw, 128)
    assert rccl._id_bytes(back) == raw
