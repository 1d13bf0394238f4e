"""`SFMLearnerLoss`: the loss half of the reference's ``SFMLearner`` link
(models/base_model.py:28-124) as a drop-in object.  DispNet / PoseNet are out of scope, so their
outputs (``pred_disps``, ``pred_poses``, ``pred_maskes``) are passed in after the four
positional arguments of the reference's ``__call__``; everything else -- constructor keys,
argument order and shapes, the returned scalar ``Variable``, ``loss.backward()``, the five
``chainer.report`` keys -- is the reference's.
"""
from __future__ import annotations

import torch

from . import ops
from .chainer_surface import Function, Variable, as_array, config, report

__all__ = ["SFMLearnerLoss", "parse_dict"]


def parse_dict(dic, key, value=None):
    """models/base_model.py:24-25"""
    return value if dic is None or key not in dic else dic[key]


class _FusedLossFunction(Function):
    """inputs: disps[0..S-1], poses[0..n-1], (masks[0..S-1]) -> total_loss (scalar).
    With backprop enabled the forward launch already produces every gradient
    (sfm_loss_fwd_bwd); backward only hands them out, scaled by the upstream gradient."""

    def __init__(self, fused, S, n, with_masks, need_grad):
        self.fused, self.S, self.n, self.with_masks, self.need_grad = fused, S, n, with_masks, need_grad

    def forward_gpu(self, inputs):
        loss5 = self.fused.forward_backward() if self.need_grad else self.fused.forward()
        self.loss5 = loss5
        return loss5[0:1].reshape(()),

    def backward_gpu(self, inputs, grad_outputs):
        gy = grad_outputs[0]
        unit = getattr(self._outputs[0], "_unit_grad", False)
        f = self.fused
        grads = list(f.d_disps) + list(f.d_poses) + (list(f.d_masks) if self.with_masks else [])
        return tuple(g if unit else g * gy for g in grads)


class SFMLearnerLoss:
    """Sfm Learner loss: multi-scale photometric (+SSIM) + smoothness + explainability."""

    def __init__(self, config, pretrained_model=None, smooth_mode="second_order"):
        # models/base_model.py:34-39
        self.n_sources = config['seq_len'] - 1
        self.smooth_reg = config['smooth_reg']
        self.exp_reg = config['exp_reg']
        self.ssim_rate = parse_dict(config, 'ssim_rate', 0.0)
        # base_model.py:75-80: the second-order form is live, the edge-aware one is commented out there
        self.smooth_mode = smooth_mode
        self.xp = torch

    def __call__(self, tgt_img, src_imgs, intrinsics, inv_intrinsics, pred_disps, pred_poses, pred_maskes=None,
                 norm_batch=None):
        """
           Args:
               tgt_img: target image. Shape is (Batch, 3, H, W)
               src_imgs: source images. Shape is (Batch, ?, 3, H, W)
               intrinsics: Shape is (Batch, ?, 3, 3)
               inv_intrinsics: unused, as in the reference (base_model.py:48)
               pred_disps: list of Variable (Batch, 1, H>>s, W>>s)   -- DispNet output (:59)
               pred_poses: list of Variable (Batch, 6)               -- PoseNet output (:62)
               pred_maskes: list of Variable (Batch, ?, H>>s, W>>s)  -- explainability logits (:62) or None
               norm_batch: global batch size when this call holds a shard of the batch
           Return:
               loss (Variable).
        """
        tgt = as_array(tgt_img)
        src = as_array(src_imgs)
        batchsize, n_sources, _, H, W = src.shape                              # :57
        stacked_src_imgs = src.reshape(batchsize, -1, H, W)                    # :58
        n_scales = len(pred_disps)                                             # :66
        do_exp = self.exp_reg is not None and self.exp_reg > 0                 # :61
        if n_sources != len(pred_poses):
            raise TypeError("src_imgs has %d sources but %d poses were given" % (n_sources, len(pred_poses)))
        # :69-72 -- curr_tgt_img / curr_src_imgs of every scale, ONE launch for both tensors, written pixel-interleaved
        # (the layout the fused loss kernels fetch with the fewest loads; values identical to the planar pyramid)
        tgt_pyr, src_pyr = ops.pyramid_pair_hwc(tgt, stacked_src_imgs, n_scales)
        fused = ops.FusedLoss(smooth_reg=self.smooth_reg or 0.0, exp_reg=self.exp_reg or 0.0,
                              ssim_rate=self.ssim_rate or 0.0, smooth_mode=self.smooth_mode)
        fused.bind(tgt_pyr, src_pyr, as_array(intrinsics), [as_array(d) for d in pred_disps],
                   [as_array(p) for p in pred_poses],
                   [as_array(m) for m in pred_maskes] if do_exp else None, norm_B=norm_batch, layout="hwc")
        inputs = list(pred_disps) + list(pred_poses) + (list(pred_maskes) if do_exp else [])
        need_grad = config.enable_backprop and any(isinstance(v, Variable) and v.requires_grad for v in inputs)
        node = _FusedLossFunction(fused, n_scales, n_sources, do_exp, need_grad)
        total_loss = node(*inputs)
        l5 = node.loss5
        report({'total_loss': l5[0]}, self)                                    # :119-123
        report({'pixel_loss': l5[1]}, self)
        report({'smooth_loss': l5[2]}, self)
        report({'exp_loss': l5[3]}, self)
        report({'ssim_loss': l5[4]}, self)
        return total_loss
