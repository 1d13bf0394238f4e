"""Importable alias of the package directory `sfm-learner-chainer_amd/` (its name, fixed by the
repository layout, is not a valid Python identifier): `import sfmwarp`."""
import importlib
import sys

_pkg = importlib.import_module("sfm-learner-chainer_amd")
sys.modules[__name__] = _pkg
