"""MI355X-native photometric view-synthesis loss path of SfM-Learner.

Layers, bottom up:
  csrc/ + libsfmwarp.so   hand-written HIP kernels for gfx950 behind the C ABI of include/sfmwarp.h
  _lib                    ctypes binding (no torch types cross it)
  ops                     array-level wrappers (torch.Tensor = device array container)
  chainer_surface         the slice of the Chainer API the reference's path is written against
  functions / links       the reference's operators and loss link under their own names
  dist                    batch sharding + RCCL all-reduce of the reported scalars
  synth                   seeded synthetic inputs (host, NumPy)

Importing `functions`, `links` or `ops` requires the built library; there is no CPU fallback.
"""
