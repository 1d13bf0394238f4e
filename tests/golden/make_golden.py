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


if __name__ == "__main__":
    make_interp()
    make_euler()
