"""The reference's operators of the view-synthesis path under their own names and call
signatures, running on the HIP kernels of libsfmwarp.so.

  spatial_transformer_sampler_interp(x, grid)      models/spational_transformer_sampler_interp.py:152-159
  SpatialTransformerSamplerInterp                  models/spational_transformer_sampler_interp.py:9-149
  spatial_transformer_sampler(x, grid)             F.spatial_transformer_sampler as called at models/transform.py:189
  proj_tgt_to_src(vec, K, N)                       models/transform.py:64-91
  projective_inverse_warp(imgs, depthes, poses, K) models/transform.py:156-193
  resize_images(x, output_shape)                   F.resize_images as called at models/base_model.py:71-72
"""
from __future__ import annotations

import torch

from . import ops
from .chainer_surface import Function, Variable, argument, as_array, type_check

__all__ = ["SpatialTransformerSamplerInterp", "spatial_transformer_sampler_interp", "SpatialTransformerSampler",
           "spatial_transformer_sampler", "ProjTgtToSrc", "proj_tgt_to_src", "ProjectiveInverseWarp",
           "projective_inverse_warp", "resize_images", "DispActivation", "disp_activation"]


def _sampler_type_check(in_types):
    # models/spational_transformer_sampler_interp.py:11-24
    n_in = in_types.size()
    type_check.expect(2 == n_in)
    x_type = in_types[0]
    grid_type = in_types[1]
    type_check.expect(
        x_type.dtype.char == 'f',
        grid_type.dtype.char == 'f',
        x_type.ndim == 4,
        grid_type.ndim == 4,
        grid_type.shape[1] == 2,
        x_type.shape[0] == grid_type.shape[0],
    )


class SpatialTransformerSamplerInterp(Function):
    """Bilinear sampler on PIXEL coordinates; exactly 0 outside [0,W-1) x [0,H-1); gx == 0."""

    def check_type_forward(self, in_types):
        _sampler_type_check(in_types)

    def forward_gpu(self, inputs):
        x, grid = inputs
        return ops.interp_fwd(x, grid),                         # _forward, :32-78

    def backward_gpu(self, inputs, grad_outputs):
        x, grid = inputs
        gy, = grad_outputs
        gx, ggrid = ops.interp_bwd(x, grid, gy, want_gx=True)   # _backward, :86-149
        return gx, ggrid


def spatial_transformer_sampler_interp(x, grid, **kwargs):
    argument.check_unexpected_kwargs(
        kwargs, use_cudnn="The argument \"use_cudnn\" is not "
        "supported anymore. "
        "Use chainer.using_config('use_cudnn', value) "
        "context where value can be `always`, `never`, or `auto`.")
    argument.assert_kwargs_empty(kwargs)
    return SpatialTransformerSamplerInterp()(x, grid)


class SpatialTransformerSampler(Function):
    """Chainer's built-in sampler: normalized grid in [-1,1], zero padding (transform.py:189)."""

    def check_type_forward(self, in_types):
        _sampler_type_check(in_types)

    def forward_gpu(self, inputs):
        x, grid = inputs
        return ops.sampler_fwd(x, grid),

    def backward_gpu(self, inputs, grad_outputs):
        x, grid = inputs
        gy, = grad_outputs
        gx, ggrid = ops.sampler_bwd(x, grid, gy, want_gx=True)
        return gx, ggrid


def spatial_transformer_sampler(x, grid, **kwargs):
    argument.check_unexpected_kwargs(
        kwargs, use_cudnn="The argument \"use_cudnn\" is not supported anymore.")
    argument.assert_kwargs_empty(kwargs)
    return SpatialTransformerSampler()(x, grid)


class ProjTgtToSrc(Function):
    """(vec (N,6), K (N,3,3)) -> projection (N,4,4), kept on the device (the reference moves
    both to the CPU and the result back, transform.py:76-80,89-90)."""

    def check_type_forward(self, in_types):
        type_check.expect(in_types.size() == 2, in_types[0].dtype.char == 'f', in_types[1].dtype.char == 'f',
                          in_types[0].ndim == 2, in_types[0].shape[1] == 6, in_types[1].ndim == 3,
                          in_types[1].shape[1:] == (3, 3), in_types[0].shape[0] == in_types[1].shape[0])

    def forward_gpu(self, inputs):
        vec, K = inputs
        return ops.pose_proj_fwd(vec, K),

    def backward_gpu(self, inputs, grad_outputs):
        vec, K = inputs
        return ops.pose_proj_bwd(vec, K, grad_outputs[0]), None


def proj_tgt_to_src(vec, K, N=None, xp=None, use_cpu=True):
    """transform.py:64-91.  `N`, `xp` and `use_cpu` are accepted for call compatibility and ignored."""
    return ProjTgtToSrc()(vec, K)


class ProjectiveInverseWarp(Function):
    """inputs: imgs (N,3,H,W), depthes (N,3,H*W), poses (N,6), K (N,3,3) -> (N,3,H,W)."""

    def check_type_forward(self, in_types):
        type_check.expect(in_types.size() == 4)
        imgs, dep, poses, K = in_types
        type_check.expect(
            imgs.dtype.char == 'f', dep.dtype.char == 'f', poses.dtype.char == 'f', K.dtype.char == 'f',
            imgs.ndim == 4, dep.ndim == 3, poses.ndim == 2, K.ndim == 3,
            dep.shape[0] == imgs.shape[0], dep.shape[1] == 3, dep.shape[2] == imgs.shape[2] * imgs.shape[3],
            poses.shape == (imgs.shape[0], 6), K.shape == (imgs.shape[0], 3, 3),
        )

    def forward_gpu(self, inputs):
        imgs, depthes, poses, K = inputs
        return ops.warp_fwd(imgs, depthes, poses, K),

    def backward_gpu(self, inputs, grad_outputs):
        imgs, depthes, poses, K = inputs
        d_depth, d_pose, d_src = ops.warp_bwd(imgs, depthes, poses, K, grad_outputs[0], want_d_src=True)
        return d_src, d_depth, d_pose, None


def projective_inverse_warp(imgs, depthes, poses, K):
    """
    Args:
        imgs(Variable or array): Source images. Shape is [N, 3, H, W]
        depthes(Variable): Predicted depthes. Shape is [N, 3, H*W]
        poses(Variable): Predicted poses. Shape is [N, 6]
        K(array): [N, 3, 3]
    Return:
        transformed images of shape [N, 3, H, W]
    """
    d = as_array(depthes)
    if isinstance(d, torch.Tensor) and d.dim() == 3 and d.stride(1) == 0:
        # F.broadcast_to result of base_model.py:83-84: materialise the three rows (values identical)
        depthes = Variable(d.contiguous(), requires_grad=False) if not isinstance(depthes, Variable) else depthes
    return ProjectiveInverseWarp()(imgs, depthes, poses, K)


def resize_images(x, output_shape):
    """F.resize_images(x, (out_H, out_W)): bilinear, align-corners; returns a Variable whose
    `.data` is what the reference takes (base_model.py:71-72)."""
    return Variable(ops.resize(as_array(x), output_shape), requires_grad=False)


class DispActivation(Function):
    """DISP_SCALING * F.sigmoid(x) + MIN_DISP for the disparity logits of all scales at once
    (models/disp_net.py:7-8,104,110,116,122): n inputs -> n outputs, one launch each way."""

    def check_type_forward(self, in_types):
        type_check.expect(in_types.size() >= 1)
        for t in in_types:
            type_check.expect(t.dtype.char == 'f')

    def forward_gpu(self, inputs):
        self._disps = ops.disp_act_fwd(list(inputs))
        return tuple(self._disps)

    def backward_gpu(self, inputs, grad_outputs):
        gs = [g if g is not None else torch.zeros_like(d) for g, d in zip(grad_outputs, self._disps)]
        return tuple(ops.disp_act_bwd(self._disps, gs))


def disp_activation(xs):
    """[10 * sigmoid(x) + 0.01 for x in xs] -> list of Variables"""
    out = DispActivation()(*xs)
    return list(out) if isinstance(out, tuple) else [out]
