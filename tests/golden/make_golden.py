#!/usr/bin/env python3
"""Generates the golden fixtures in this directory.  Run in the BUILD container only
(`python tests/golden/make_golden.py`): it needs /root/reference, which does not exist on
the GPU box.  The fixtures are data (inputs + expected outputs); no reference source is
copied.

1. interp_sampler_*.npz -- inputs and outputs of the reference's own
   ``SpatialTransformerSamplerInterp._forward/_backward``
   (/root/reference/models/spational_transformer_sampler_interp.py:32-149), executed
   unmodified.  That file imports ``chainer`` only for the ``Function`` base class,
   ``cuda.get_array_module`` and two kwarg checkers; Chainer is not installed and cannot be
   (no network), so five NAMES are provided in ``sys.modules`` for the duration of the import
   (an empty ``Function`` base class, ``get_array_module -> numpy`` and empty ``argument`` /
   ``type_check`` modules).  None of them contains arithmetic: every number in the fixture
   is computed by the reference's own NumPy code.

2. euler_odom_util.npz -- rotation matrices from the reference's plain-NumPy
   ``kitti_eval/odom_util.py:167-200 euler2mat(z, y, x)``, which composes the same
   X.Y.Z product as ``models/transform.py:11-40``.

3. intrinsics_aug.npz -- the intrinsics path of the data pipeline (round 6): ``make_intrinsics_matrix``,
   ``data_augmentation`` (K updates, np.random order, crop offsets, flip decision, crop + flip indexing) and
   ``get_multi_scale_intrinsics`` of ``datasets/kitti/kitti_raw_transformed.py:17-93``, executed unmodified with
   arithmetic-free stand-ins for its missing imports (see _load_reference_transform).
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load_reference_interp():
    chainer = types.ModuleType("chainer")
    function = types.ModuleType("chainer.function")
    utils = types.ModuleType("chainer.utils")
    argument = types.ModuleType("chainer.utils.argument")
    type_check = types.ModuleType("chainer.utils.type_check")
    cuda = types.ModuleType("chainer.cuda")

    class Function(object):
        pass

    function.Function = Function
    cuda.get_array_module = lambda *a: np
    chainer.function, chainer.utils, chainer.cuda = function, utils, cuda
    utils.argument, utils.type_check = argument, type_check
    names = {"chainer": chainer, "chainer.function": function, "chainer.utils": utils,
             "chainer.utils.argument": argument, "chainer.utils.type_check": type_check,
             "chainer.cuda": cuda}
    saved = {k: sys.modules.get(k) for k in names}
    sys.modules.update(names)
    try:
        spec = importlib.util.spec_from_file_location(
            "_ref_interp", os.path.join(REF, "models", "spational_transformer_sampler_interp.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def make_interp():
    mod = _load_reference_interp()
    f = mod.SpatialTransformerSamplerInterp()
    cases = {
        # name: (B, C, H, W, oH, oW, kind)
        "small": (2, 3, 8, 13, 8, 13, "jitter"),
        "ragged": (3, 3, 5, 7, 4, 9, "uniform"),
        "c1": (1, 1, 6, 6, 6, 6, "integer"),
        "c5": (2, 5, 9, 11, 3, 17, "uniform"),
        "border": (1, 3, 4, 6, 1, 12, "border"),
        "kitti_s3": (2, 3, 16, 52, 16, 52, "jitter"),
    }
    for name, (B, C, H, W, oH, oW, kind) in cases.items():
        rng = np.random.RandomState(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
        x = rng.uniform(-1, 1, size=(B, C, H, W)).astype(np.float32)
        if kind == "jitter":
            ys, xs = np.meshgrid(np.arange(oH), np.arange(oW), indexing="ij")
            base = np.stack([xs * (W - 1) / max(oW - 1, 1), ys * (H - 1) / max(oH - 1, 1)])[None]
            grid = (base + rng.normal(0, 1.5, size=(B, 2, oH, oW))).astype(np.float32)
        elif kind == "uniform":
            grid = np.stack([rng.uniform(-2, W + 1, size=(B, oH, oW)),
                             rng.uniform(-2, H + 1, size=(B, oH, oW))], axis=1).astype(np.float32)
        elif kind == "integer":
            ys, xs = np.meshgrid(np.arange(oH), np.arange(oW), indexing="ij")
            grid = np.stack([xs, ys])[None].repeat(B, 0).astype(np.float32)
        elif kind == "border":
            us = np.array([-1.5, -0.5, 0.0, 0.5, W - 2, W - 1.5, W - 1 - 1e-3, W - 1, W - 0.5, W + 0.5, 2.25, 1.0],
                          dtype=np.float32)
            grid = np.stack([us, np.linspace(-0.5, H - 0.5, 12).astype(np.float32)])[None].reshape(1, 2, 1, 12)
        gy = rng.uniform(-1, 1, size=(B, C, oH, oW)).astype(np.float32)
        y, = f._forward((x, grid))
        gx, ggrid = f._backward((x, grid), (gy,))
        np.savez_compressed(os.path.join(HERE, "interp_sampler_%s.npz" % name),
                            x=x, grid=grid, gy=gy, y=np.ascontiguousarray(y),
                            gx=np.ascontiguousarray(gx), ggrid=np.ascontiguousarray(ggrid))
        print("interp_sampler_%s: y %s %s  ggrid %s %s  |gx|max %g" % (
            name, y.shape, y.dtype, ggrid.shape, ggrid.dtype, float(np.abs(gx).max())))


def make_euler():
    sys.path.insert(0, REF)
    try:
        spec = importlib.util.spec_from_file_location("_ref_odom_util", os.path.join(REF, "kitti_eval", "odom_util.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.pop(0)
    rng = np.random.RandomState(7)
    r = rng.uniform(-np.pi, np.pi, size=(64, 3))
    r[:8] = rng.uniform(-0.05, 0.05, size=(8, 3))            # pose-net sized angles
    r[8] = [0.3, 0.0, 0.0]
    r[9] = [0.0, -0.7, 0.0]
    r[10] = [0.0, 0.0, 1.1]
    r[11] = [0.0, 0.0, 0.0]
    R = np.stack([mod.euler2mat(z=float(a[2]), y=float(a[1]), x=float(a[0])) for a in r])
    np.savez_compressed(os.path.join(HERE, "euler_odom_util.npz"), r_xyz=r, R=R)
    print("euler_odom_util:", R.shape, R.dtype)


def _load_reference_transform(record):
    """datasets/kitti/kitti_raw_transformed.py imported unmodified.  Its module-level imports need five names this image lacks (cv2,
    chainer.datasets.TransformDataset, chainer.functions, datasets.kitti.kitti_raw_dataset.KittiRawDataset): arithmetic-free
    stand-ins for the duration of the import.  The ONE Chainer call inside data_augmentation, F.resize_images (:39), is replaced by
    a stub that returns an array of the REQUESTED SHAPE whose values are the index of each element, (frame * 512 + row) * 2048 + col
    -- no resampling arithmetic -- and notes the requested size in `record`: everything the fixture holds (intrinsics, crop
    offsets, flip decision, the crop / flip INDEXING) is computed by the reference's own NumPy statements, none of which depends on
    the resampled pixel values."""
    chainer = types.ModuleType("chainer")
    ch_datasets = types.ModuleType("chainer.datasets")
    ch_functions = types.ModuleType("chainer.functions")
    cv2 = types.ModuleType("cv2")
    ds = types.ModuleType("datasets")
    ds_kitti = types.ModuleType("datasets.kitti")
    ds_raw = types.ModuleType("datasets.kitti.kitti_raw_dataset")

    class TransformDataset(object):
        def __init__(self, *a, **k):
            pass

    class KittiRawDataset(object):
        pass

    class _Var(object):
        def __init__(self, data):
            self.data = data

    def resize_images(imgs, size):
        out_h, out_w = int(size[0]), int(size[1])
        n, c = imgs.shape[:2]
        record.append((out_h, out_w))
        f = np.arange(n, dtype=np.float64)[:, None, None, None]
        y = np.arange(out_h, dtype=np.float64)[None, None, :, None]
        x = np.arange(out_w, dtype=np.float64)[None, None, None, :]
        return _Var(np.broadcast_to((f * 512 + y) * 2048 + x, (n, c, out_h, out_w)).astype(np.float32))

    ch_datasets.TransformDataset = TransformDataset
    ch_functions.resize_images = resize_images
    chainer.datasets, chainer.functions = ch_datasets, ch_functions
    ds.kitti = ds_kitti
    ds_kitti.kitti_raw_dataset = ds_raw
    ds_raw.KittiRawDataset = KittiRawDataset
    names = {"chainer": chainer, "chainer.datasets": ch_datasets, "chainer.functions": ch_functions, "cv2": cv2,
             "datasets": ds, "datasets.kitti": ds_kitti, "datasets.kitti.kitti_raw_dataset": ds_raw}
    saved = {k: sys.modules.get(k) for k in names}
    sys.modules.update(names)
    try:
        spec = importlib.util.spec_from_file_location(
            "_ref_kitti_raw_transformed", os.path.join(REF, "datasets", "kitti", "kitti_raw_transformed.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def make_intrinsics():
    """intrinsics_aug.npz -- the K path of the reference's data pipeline, datasets/kitti/kitti_raw_transformed.py:17-93, for seeded
    np.random states: make_intrinsics_matrix (:17-21), data_augmentation's intrinsics / crop offsets / flip decision and the crop +
    flip indexing on an index-valued image (:23-74), get_multi_scale_intrinsics (:76-93) and _transform (:95-102)."""
    record = []
    mod = _load_reference_transform(record)
    cases = [
        # seed, H, W, n_src, n_scales, (fx, fy, cx, cy)
        (1, 128, 416, 2, 4, (241.674463, 246.284868, 204.168010, 59.000832)),
        (6, 128, 416, 2, 4, (235.1, 250.9, 210.5, 63.25)),
        (14, 256, 832, 4, 4, (483.348926, 492.569736, 408.336020, 118.001664)),
        (5, 128, 416, 4, 4, (241.674463, 246.284868, 204.168010, 59.000832)),
        (8, 96, 320, 2, 3, (190.0, 195.5, 160.25, 47.75)),
        (20, 128, 416, 2, 4, (250.0, 250.0, 208.0, 64.0)),
        (21, 37, 70, 1, 2, (60.3, 61.7, 34.2, 18.9)),
        (34, 128, 416, 2, 4, (241.674463, 246.284868, 204.168010, 59.000832)),
    ]
    out = {"n_cases": np.int64(len(cases))}
    for k, (seed, H, W, n_src, n_scales, (fx, fy, cx, cy)) in enumerate(cases):
        K_in = mod.make_intrinsics_matrix(fx, fy, cx, cy)                            # :17-21
        tgt = np.zeros((3, H, W), dtype=np.float32)
        src = np.zeros((n_src, 3, H, W), dtype=np.float32)
        del record[:]
        np.random.seed(seed)
        t, s, K_out = mod.data_augmentation(tgt, src, K_in.copy())                   # :23-74
        K_multi = np.stack(mod.get_multi_scale_intrinsics(K_out, n_scales))          # :76-93
        (sh, sw), = record
        first = int(t[0, 0, 0])                                                      # index of the output's top-left element
        last = int(t[0, 0, W - 1])
        oy, c0, c1 = first // 2048, first % 2048, last % 2048
        flip = c1 < c0
        ox = c1 if flip else c0
        # the same draws through _transform (:95-102): data_augmentation + get_multi_scale_intrinsics with the hard-coded 4 scales
        np.random.seed(seed)
        tt, ss, Kt, Kt2 = mod._transform((tgt, src, K_in.copy(), None), n_scale=n_scales)
        assert np.array_equal(tt, t) and np.array_equal(np.stack(Kt), np.stack(mod.get_multi_scale_intrinsics(K_out, len(Kt))))
        pre = "c%d_" % k
        out.update({pre + "seed": np.int64(seed), pre + "hw": np.array([H, W], np.int64), pre + "n_src": np.int64(n_src),
                    pre + "K_in": K_in, pre + "K_out": np.asarray(K_out), pre + "K_multi": K_multi,
                    pre + "scaled_hw": np.array([sh, sw], np.int64), pre + "offset_yx": np.array([oy, ox], np.int64),
                    pre + "flip": np.bool_(flip),
                    # the index-valued output (value = (frame * 512 + row) * 2048 + col of the resized stack): border rows / columns of
                    # the target (every channel holds the same indices) and the four corners of every source frame
                    pre + "tgt_rows": np.stack([t[0, 0], t[0, H // 2], t[0, H - 1]]).astype(np.int32),
                    pre + "tgt_cols": np.stack([t[0, :, 0], t[0, :, W // 2], t[0, :, W - 1]]).astype(np.int32),
                    pre + "tgt_channels_equal": np.bool_(np.array_equal(t[0], t[1]) and np.array_equal(t[0], t[2])),
                    pre + "src_corners": np.stack([[f[0, 0, 0], f[0, 0, W - 1], f[0, H - 1, 0], f[0, H - 1, W - 1]] for f in s]).astype(np.int32)})
        print("intrinsics_aug case %d: seed %d %dx%d -> scaled %dx%d offset (%d, %d) flip %s  K_out fx %.4f cx %.4f" % (
            k, seed, H, W, sh, sw, oy, ox, flip, K_out[0, 0], K_out[0, 2]))
    np.savez_compressed(os.path.join(HERE, "intrinsics_aug.npz"), **out)


if __name__ == "__main__":
    make_interp()
    make_euler()
    make_intrinsics()
