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

    def __init__(self, fused, S, n, with_masks, need_grad, run=None, state=None, frames=None):
        self.fused, self.S, self.n, self.with_masks, self.need_grad, self.run = fused, S, n, with_masks, need_grad, run
        # (tgt, stacked sources) at full resolution: pyramids + loss in ONE call through the C ABI (sfm_step_fwd_bwd); None: the
        # pyramids of this call have been built already (planar layout, HIP-graph replay)
        self.frames = frames
        # the call of the link this node belongs to: the gradient arrays of a cached FusedLoss are valid for its latest call only
        self.state, self.generation = state, (state.calls if state is not None else 0)

    def forward_gpu(self, inputs):
        # The five scalars of THIS call get an array of their own (as every Chainer Function output does): the returned loss and
        # the reported values stay what they were when a later call reuses the link's cached buffers.
        if self.run is not None:
            loss5 = self.run().clone()         # pyramid + fused launch of this call, replayed from a HIP graph (fixed output)
        else:
            out = torch.empty((5,), dtype=torch.float32, device=self.fused.device)
            if self.frames is not None:
                loss5 = self.fused.step_from_frames(self.frames[0], self.frames[1], grad=self.need_grad, out=out)
            else:
                loss5 = self.fused.forward_backward(out=out) if self.need_grad else self.fused.forward(out=out)
        self.loss5 = loss5
        return loss5[0:1].reshape(()),

    def backward_gpu(self, inputs, grad_outputs):
        if self.state is not None and self.state.calls != self.generation:
            raise RuntimeError(
                "SFMLearnerLoss(cache_buffers=True): this loss belongs to call %d of the link, but the link has been called "
                "again since (call %d) and its cached gradient buffers now hold the later call's gradients.  Call backward() "
                "before the next forward, or construct the link with cache_buffers=False." % (self.generation, self.state.calls))
        gy = grad_outputs[0]
        # the seed of ones that Variable.backward() creates itself is not multiplied in (flag valid during that sweep only);
        # any gradient the caller has set on the loss (e.g. a loss scale) is
        unit = getattr(self._outputs[0], "_unit_grad", False)
        f = self.fused
        grads = list(f.d_disps) + list(f.d_poses) + (list(f.d_masks) if self.with_masks else [])
        return tuple(g if unit else g * gy for g in grads)


class _Cached:
    """What one link keeps between calls for one set of shapes."""
    __slots__ = ("fused", "pyr", "layout", "graph", "graph_key", "graph_stream", "calls")


class _Repeat:
    """The arguments of the link's previous call, for the fast path of `SFMLearnerLoss.__call__`: a call that passes the very same
    objects, holding the very same arrays at the same addresses, has nothing to validate, reshape or re-bind."""
    __slots__ = ("objs", "tensors", "ptrs", "st", "tgt", "stacked", "inputs", "n_scales", "n_sources", "do_exp", "norm_batch")


# SFM_LAYOUT_HWC forms the byte offset of a gather inside one image exactly in fp32 (include/sfmwarp.h): an image of a scale
# must have fewer than 2^24 / 12 pixels there.  Larger frames take the reference's planar layout (same results).
HWC_MAX_PIXELS = (1 << 24) // 12


def _build_pyramids(st, tgt, stacked, n_scales):
    """models/base_model.py:69-72: curr_tgt_img / curr_src_imgs of every scale into the buffers the bound loss reads."""
    if st.layout == "hwc":
        # ONE launch for both tensors, written pixel-interleaved (the layout the fused loss kernels fetch with the fewest loads;
        # values identical to the planar pyramid)
        st.pyr = ops.pyramid_pair_hwc(tgt, stacked, n_scales, out=st.pyr)
        return
    # planar: scales 1.. in one launch per tensor; scale 0 is the frame itself, copied into the bound buffer when the caller's
    # array is not the one the descriptor was bound to
    if st.pyr is None:
        st.pyr = (ops.pyramid(tgt.clone(), n_scales), ops.pyramid(stacked.clone(), n_scales))
        return
    for x, pyr in ((tgt, st.pyr[0]), (stacked, st.pyr[1])):
        if x.data_ptr() != pyr[0].data_ptr():
            pyr[0].copy_(x)
        ops.pyramid(pyr[0], n_scales, out=pyr)


class SFMLearnerLoss:
    """Sfm Learner loss: multi-scale photometric (+SSIM) + smoothness + explainability.

    cache_buffers (default True): the image pyramids, the workspace and the gradient arrays are allocated once per set
      of input shapes and reused by every later call (only the input pointers are re-bound).  The arrays that
      `loss.backward()` hands to `x.grad` are therefore owned by the link and are overwritten by its next call with the same
      shapes -- the cleargrads() / forward / backward / update cycle of the reference's trainer (a caller that keeps
      gradients across iterations copies them, or passes cache_buffers=False).  A `backward()` on the loss of an EARLIER
      call, after the link has been called again, raises instead of handing out the later call's gradients.  The returned
      loss and the five reported scalars are per-call arrays: they keep their values.
    projection (default "fast"): "reference_order" evaluates the per-pixel projection in the reference's own rounding sequence
      (warped pixels within 1e-4 of the reference's at any frame size; the launch is about 10 % longer; include/sfmwarp.h).
    use_graph (default False): when a call repeats the previous call's arrays exactly (same addresses: static input buffers),
      the pyramid launch and the three launches of the loss are replayed from one HIP graph.
    """

    def __init__(self, config, pretrained_model=None, smooth_mode="second_order", cache_buffers=True, use_graph=False,
                 projection="fast"):
        # models/base_model.py:34-39
        self.n_sources = config['seq_len'] - 1
        self.smooth_reg = config['smooth_reg']
        self.exp_reg = config['exp_reg']
        self.ssim_rate = parse_dict(config, 'ssim_rate', 0.0)
        # base_model.py:75-80: the second-order form is live, the edge-aware one is commented out there
        self.smooth_mode = smooth_mode
        # "fast" or "reference_order": how the kernels evaluate transform.py:94-133 per pixel (include/sfmwarp.h, SFM_PROJECTION_*)
        self.projection = projection
        self.xp = torch
        self.cache_buffers = cache_buffers
        self.use_graph = use_graph
        self._cache = {}
        self._repeat = None

    def _state(self, tgt, stacked, intrinsics, disps, poses, masks, norm_batch):
        """The bound FusedLoss + pyramid buffers for these shapes (built on first use)."""
        key = (tgt.device, tuple(tgt.shape), tuple(stacked.shape), tuple(intrinsics.shape), tuple(tuple(a.shape) for a in disps),
               masks is not None, norm_batch)
        st = self._cache.get(key) if self.cache_buffers else None
        if st is None:
            st = _Cached()
            st.layout = "hwc" if tgt.shape[2] * tgt.shape[3] < HWC_MAX_PIXELS else "planar"
            st.pyr = None
            _build_pyramids(st, tgt, stacked, len(disps))
            st.fused = ops.FusedLoss(smooth_reg=self.smooth_reg or 0.0, exp_reg=self.exp_reg or 0.0,
                                     ssim_rate=self.ssim_rate or 0.0, smooth_mode=self.smooth_mode, projection=self.projection)
            st.fused.bind(st.pyr[0], st.pyr[1], intrinsics, disps, poses, masks, norm_B=norm_batch, layout=st.layout)
            st.graph = st.graph_key = st.graph_stream = None
            st.calls = 0
            if self.cache_buffers:
                self._cache.clear()            # one set of shapes at a time: a training loop has one
                self._cache[key] = st
            return st, True
        st.fused.rebind(intrinsics, disps, poses, masks)
        return st, False

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
        # Fast path (round 6; the reference trains at B = 4, where a step is 25 us of GPU work and the host side of this call decides
        # the step time): the previous call's objects again, holding the same arrays at the same addresses -- static input buffers --
        # are neither validated nor re-bound a second time.
        rp = self._repeat
        if rp is not None and norm_batch == rp.norm_batch:
            cur = [tgt_img, src_imgs, intrinsics]
            cur += pred_disps
            cur += pred_poses
            if rp.do_exp and pred_maskes is not None:
                cur += pred_maskes
            same = len(cur) == len(rp.objs)
            if same:
                for a, b, t, p in zip(cur, rp.objs, rp.tensors, rp.ptrs):
                    d = a.data if type(a) is Variable else a
                    if a is not b or d is not t or d.data_ptr() != p:
                        same = False
                        break
            if same:
                return self._finish(rp.st, rp.inputs, rp.n_scales, rp.n_sources, rp.do_exp, None, (rp.tgt, rp.stacked))
        tgt = ops._dev(as_array(tgt_img), "tgt_img", 4)
        src = as_array(src_imgs)
        batchsize, n_sources, _, H, W = src.shape                              # :57
        stacked_src_imgs = ops._dev(src.reshape(batchsize, -1, H, W), "src_imgs", 4)   # :58
        n_scales = len(pred_disps)                                             # :66
        do_exp = self.exp_reg is not None and self.exp_reg > 0                 # :61
        if n_sources != len(pred_poses):
            raise TypeError("src_imgs has %d sources but %d poses were given" % (n_sources, len(pred_poses)))
        K = as_array(intrinsics)
        disps = [as_array(d) for d in pred_disps]
        poses = [as_array(p) for p in pred_poses]
        masks = [as_array(m) for m in pred_maskes] if do_exp else None
        inputs = list(pred_disps) + list(pred_poses) + (list(pred_maskes) if do_exp else [])
        need_grad = config.enable_backprop and any(isinstance(v, Variable) and v.requires_grad for v in inputs)
        st, fresh = self._state(tgt, stacked_src_imgs, K, disps, poses, masks, norm_batch)
        run = None
        if self.use_graph and self.cache_buffers:
            run = self._graph_step(st, tgt, stacked_src_imgs, n_scales, need_grad, fresh)
        frames = None
        if run is None:
            if st.layout == "hwc":
                frames = (tgt, stacked_src_imgs)          # pyramids + loss in one call (sfm_step_fwd_bwd)
            elif not fresh:
                _build_pyramids(st, tgt, stacked_src_imgs, n_scales)                # :69-72
        self._repeat = None
        if self.cache_buffers and frames is not None and not self.use_graph:
            rp = _Repeat()
            rp.objs = [tgt_img, src_imgs, intrinsics] + list(pred_disps) + list(pred_poses) + (list(pred_maskes) if do_exp else [])
            rp.tensors = [as_array(o) for o in rp.objs]
            rp.ptrs = [t.data_ptr() for t in rp.tensors]
            # (the arrays the kernels read are the validated, contiguous ones the descriptor is bound to; an input that had to be
            #  copied -- non-contiguous -- is not what is bound and cannot repeat)
            keep = st.fused._keep
            bound = [keep[2]] + list(keep[3]) + list(keep[4]) + (list(keep[5]) if keep[5] is not None else [])
            if tgt is rp.tensors[0] and len(bound) == len(rp.tensors) - 2 and all(a is b for a, b in zip(rp.tensors[2:], bound)) \
                    and stacked_src_imgs.data_ptr() == rp.ptrs[1]:
                rp.st, rp.tgt, rp.stacked, rp.inputs = st, tgt, stacked_src_imgs, inputs
                rp.n_scales, rp.n_sources, rp.do_exp, rp.norm_batch = n_scales, n_sources, do_exp, norm_batch
                self._repeat = rp
        return self._finish(st, inputs, n_scales, n_sources, do_exp, run, frames, need_grad)

    def _finish(self, st, inputs, n_scales, n_sources, do_exp, run, frames, need_grad=None):
        """The Function node of this call and the five reported scalars (models/base_model.py:117-124)."""
        if need_grad is None:
            need_grad = config.enable_backprop and any(isinstance(v, Variable) and v.requires_grad for v in inputs)
        st.calls += 1
        node = _FusedLossFunction(st.fused, n_scales, n_sources, do_exp, need_grad, run, st if self.cache_buffers else None, frames)
        total_loss = node(*inputs)
        l5 = node.loss5.unbind(0)
        report({'total_loss': l5[0]}, self)                                    # :119-123
        report({'pixel_loss': l5[1]}, self)
        report({'smooth_loss': l5[2]}, self)
        report({'exp_loss': l5[3]}, self)
        report({'ssim_loss': l5[4]}, self)
        return total_loss

    def _graph_step(self, st, tgt, stacked, n_scales, need_grad, fresh):
        """Returns a callable that replays [pyramid, fused loss] of THIS call from a HIP graph, or None (run eagerly).
        The graph is captured on the second call that repeats the same addresses and is dropped when they change."""
        f = st.fused
        key = (tgt.data_ptr(), stacked.data_ptr(), need_grad, bytes(f.desc))
        if st.graph is not None and st.graph_key == key:
            g = st.graph
            return lambda: (g.replay(), f.loss5)[1]
        st.graph = None
        if fresh or st.graph_key != key:
            st.graph_key = key             # first sighting of these addresses: run eagerly, capture next time
            return None
        side = torch.cuda.Stream(device=tgt.device)
        side.wait_stream(torch.cuda.current_stream(tgt.device))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            _build_pyramids(st, tgt, stacked, n_scales)
            f.forward_backward() if need_grad else f.forward()
        torch.cuda.current_stream(tgt.device).wait_stream(side)
        st.graph = g
        return lambda: (g.replay(), f.loss5)[1]
