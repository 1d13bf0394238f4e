"""The slice of the Chainer 4 API surface that the reference's hot path is written against,
re-created without Chainer (it is not installable here): ``Variable``, the old-style
``Function`` protocol (models/spational_transformer_sampler_interp.py:9-30,80-84,152-159),
``type_check.expect``, ``argument.check_unexpected_kwargs / assert_kwargs_empty``,
``report`` (models/base_model.py:119-123) and ``no_backprop_mode``
(models/base_model.py:190-191).

Arrays are ``torch.Tensor`` on a ROCm device where the reference has ``cupy.ndarray``; a
``Function`` dispatches to ``forward_gpu`` / ``backward_gpu`` on them.  There is no CPU
implementation behind ``forward_cpu``: it raises (the product path has no CPU fallback).

Only what the path needs is implemented: reverse-mode differentiation over a DAG of
``Function`` nodes, gradients accumulated into ``Variable.grad``.
"""
from __future__ import annotations

import contextlib
import threading
import weakref

import torch

__all__ = ["Variable", "Function", "InvalidType", "type_check", "argument", "report", "get_report", "clear_report",
           "no_backprop_mode", "using_config", "config", "as_array"]


# --------------------------------------------------------------------------------------------
# configuration (chainer.config / chainer.using_config / chainer.no_backprop_mode)
# --------------------------------------------------------------------------------------------
class _Config(threading.local):
    enable_backprop = True
    train = True
    type_check = True


config = _Config()


@contextlib.contextmanager
def using_config(name, value):
    if not hasattr(config, name):
        raise AttributeError("unknown config entry %r" % name)
    old = getattr(config, name)
    setattr(config, name, value)
    try:
        yield
    finally:
        setattr(config, name, old)


def no_backprop_mode():
    """chainer.function.no_backprop_mode (models/base_model.py:191)."""
    return using_config("enable_backprop", False)


# --------------------------------------------------------------------------------------------
# chainer.utils.type_check / chainer.utils.argument
# --------------------------------------------------------------------------------------------
class InvalidType(TypeError):
    """chainer.utils.type_check.InvalidType"""


_DTYPE_CHARS = {torch.float32: "f", torch.float64: "d", torch.float16: "e", torch.int32: "i", torch.int64: "l", torch.bool: "?"}


class _DType:
    def __init__(self, t):
        self._t = t
        self.char = _DTYPE_CHARS.get(t, "O")
        self.kind = "f" if t.is_floating_point else "i"

    def __eq__(self, other):
        return other is self._t or getattr(other, "_t", None) is self._t or other == self.char

    __hash__ = None


_DTYPES = {}     # one _DType per torch dtype: a Function call builds a _TypeInfo per input, every call (the link: six per step)


class _TypeInfo:
    __slots__ = ("shape", "ndim", "dtype", "name")

    def __init__(self, a, name):
        self.shape = tuple(a.shape)
        self.ndim = a.dim()
        dt = _DTYPES.get(a.dtype)
        if dt is None:
            dt = _DTYPES[a.dtype] = _DType(a.dtype)
        self.dtype = dt
        self.name = name


class _TypeInfoTuple(tuple):
    def size(self):
        return len(self)


class _TypeCheck:
    InvalidType = InvalidType

    @staticmethod
    def expect(*conditions):
        for k, c in enumerate(conditions):
            if not bool(c):
                raise InvalidType("type_check.expect: condition #%d does not hold" % k)


class _Argument:
    @staticmethod
    def check_unexpected_kwargs(kwargs, **unexpected):
        for key, message in unexpected.items():
            if key in kwargs:
                raise ValueError(message)

    @staticmethod
    def assert_kwargs_empty(kwargs):
        if kwargs:
            raise TypeError("got unexpected keyword argument(s) %s" % ", ".join("'%s'" % k for k in kwargs))


type_check = _TypeCheck()
argument = _Argument()


# --------------------------------------------------------------------------------------------
# chainer.report
# --------------------------------------------------------------------------------------------
class _Reported(threading.local):
    """Latest observations of the calling thread, per observer (Chainer scopes a report by the reporter that is current
    in the calling thread and prefixes the key with the observer's name; here the observer object itself is the scope).
    Nothing is shared between threads or between two links; an observer's entries go away with the observer."""

    def __init__(self):
        self.by_observer = weakref.WeakKeyDictionary()
        self.anonymous = {}

    def slot(self, observer, create):
        if observer is None:
            return self.anonymous
        # (a link reports five values per call, one report() each, as the reference does: the slot of the observer that reported
        #  last is remembered -- through a weak reference, so that it keeps nothing alive)
        last = self.__dict__.get("_last")
        if last is not None and last[0]() is observer:
            return last[1]
        d = self._slot(observer, create)
        if create and d is not self.anonymous:
            try:
                self.__dict__["_last"] = (weakref.ref(observer), d)
            except TypeError:
                pass
        return d

    def _slot(self, observer, create):
        try:
            if create:
                return self.by_observer.setdefault(observer, {})
            return self.by_observer.get(observer, {})
        except TypeError:      # an observer that cannot be weakly referenced
            return self.anonymous


_reported = _Reported()


def report(values, observer=None):
    """chainer.report(values, observer) (models/base_model.py:119-123): keeps the latest observation under the same keys."""
    _reported.slot(observer, True).update(values)


def get_report(observer=None):
    """The observations `observer` reported from this thread; without an observer, all of this thread's, merged."""
    if observer is not None:
        return dict(_reported.slot(observer, False))
    out = dict(_reported.anonymous)
    for d in list(_reported.by_observer.values()):
        out.update(d)
    return out


def clear_report(observer=None):
    if observer is None:
        _reported.by_observer.clear()
        _reported.anonymous.clear()
    else:
        _reported.slot(observer, False).clear()


# --------------------------------------------------------------------------------------------
# Variable / Function
# --------------------------------------------------------------------------------------------
def as_array(x):
    return x.data if isinstance(x, Variable) else x


_ONES = {}      # (device, dtype) -> the scalar 1 on that device


def _seed_of_ones(data):
    """The gradient a scalar output's backward() starts from when none has been set (Chainer: ones).  For a 0-d output it is ONE
    constant array per (device, dtype), made once: torch.ones_like launches a fill kernel, 6 us of host time and a fourth launch on
    a step that is 25 us of GPU work at the reference's batch size.  It is handed out as `loss.grad`, as in Chainer -- read it,
    do not write into it (assign `loss.grad = ...` to set a loss scale: that array is yours)."""
    if data.dim() != 0:
        return torch.ones_like(data)
    key = (data.device, data.dtype)
    one = _ONES.get(key)
    if one is None:
        one = _ONES[key] = torch.ones((), dtype=data.dtype, device=data.device)
    return one


class Variable:
    """chainer.Variable: ``.data`` (array), ``.grad``, ``.creator``, ``.backward()``."""

    def __init__(self, data, requires_grad=True, name=None):
        if not isinstance(data, torch.Tensor):
            raise TypeError("Variable wraps a torch.Tensor (device array), got %s" % type(data).__name__)
        self.data = data
        self.grad = None
        self.creator = None
        self.rank = 0
        self.requires_grad = requires_grad
        self.name = name
        self._out_index = 0
        self._unit_grad = False

    array = property(lambda self: self.data)
    shape = property(lambda self: tuple(self.data.shape))
    dtype = property(lambda self: self.data.dtype)
    ndim = property(lambda self: self.data.dim())

    def cleargrad(self):
        self.grad = None

    def __float__(self):
        return float(self.data)

    def __repr__(self):
        return "variable(%r)" % (self.data,)

    def backward(self, retain_grad=False):
        """Reverse-mode sweep from this variable.  As in Chainer, a scalar output starts from
        a gradient of one when none has been set."""
        if self.creator is None:
            return
        seeded_here = False
        if self.grad is None:
            if self.data.numel() != 1:
                raise RuntimeError("backward() on a non-scalar Variable needs .grad to be set first")
            self.grad = _seed_of_ones(self.data)
            seeded_here = True
        # lets a fused loss node skip the multiplication by one -- only during THIS sweep, and only when the seed of ones
        # was created here (a gradient the caller has set, e.g. a loss scale, is always multiplied in)
        self._unit_grad = seeded_here
        try:
            self._sweep(retain_grad)
        finally:
            self._unit_grad = False

    def _sweep(self, retain_grad):
        # topological order by rank (rank = 1 + max rank of the inputs)
        funcs, seen = [], set()

        def add(f):
            if f is not None and id(f) not in seen:
                seen.add(id(f))
                funcs.append(f)

        add(self.creator)
        while funcs:
            funcs.sort(key=lambda f: f.rank)
            f = funcs.pop()
            gys = tuple(o.grad if o is not None else None for o in f._outputs)
            if all(g is None for g in gys):
                continue
            in_data = tuple(as_array(x) for x in f._inputs)
            gxs = f.backward(in_data, gys)
            if not isinstance(gxs, tuple):
                gxs = tuple(gxs)
            for x, gx in zip(f._inputs, gxs):
                if gx is None or not isinstance(x, Variable) or not x.requires_grad:
                    continue
                x.grad = gx if x.grad is None else x.grad + gx
                add(x.creator)
            if not retain_grad:
                for o in f._outputs:
                    if o is not None and o is not self:
                        o.grad = None


class Function:
    """Old-style chainer.function.Function:

        check_type_forward(in_types); forward_cpu/forward_gpu(inputs) -> tuple;
        backward_cpu/backward_gpu(inputs, grad_outputs) -> tuple (one entry per input)

    ``inputs`` are raw arrays, not Variables (spational_transformer_sampler_interp.py:32-33)."""

    rank = 0

    def check_type_forward(self, in_types):
        pass

    def forward_cpu(self, inputs):
        raise NotImplementedError(
            "%s: CPU arrays are not supported -- this is the MI355X build, there is no CPU fallback" % type(self).__name__)

    def forward_gpu(self, inputs):
        raise NotImplementedError

    def backward_cpu(self, inputs, grad_outputs):
        raise NotImplementedError(
            "%s: CPU arrays are not supported -- this is the MI355X build, there is no CPU fallback" % type(self).__name__)

    def backward_gpu(self, inputs, grad_outputs):
        return tuple(None for _ in inputs)

    def forward(self, inputs):
        if any(isinstance(a, torch.Tensor) and a.is_cuda for a in inputs):
            return self.forward_gpu(inputs)
        return self.forward_cpu(inputs)

    def backward(self, inputs, grad_outputs):
        if any(isinstance(a, torch.Tensor) and a.is_cuda for a in inputs):
            return self.backward_gpu(inputs, grad_outputs)
        return self.backward_cpu(inputs, grad_outputs)

    def __call__(self, *inputs):
        in_data = tuple(as_array(x) for x in inputs)
        for k, a in enumerate(in_data):
            if not isinstance(a, torch.Tensor):
                raise TypeError("input %d of %s is %s, expected a Variable or a device array" % (k, type(self).__name__, type(a).__name__))
        if config.type_check and type(self).check_type_forward is not Function.check_type_forward:     # (the base's is a no-op)
            self.check_type_forward(_TypeInfoTuple(_TypeInfo(a, "in_types[%d]" % k) for k, a in enumerate(in_data)))
        outputs = self.forward(in_data)
        if not isinstance(outputs, tuple):
            raise TypeError("forward must return a tuple")
        need_graph = config.enable_backprop and any(isinstance(x, Variable) and x.requires_grad for x in inputs)
        outs = []
        for k, y in enumerate(outputs):
            v = Variable(y, requires_grad=need_graph)
            v._out_index = k
            outs.append(v)
        if need_graph:
            self._inputs = inputs
            self._outputs = outs
            self.rank = 1 + max([x.rank for x in inputs if isinstance(x, Variable)] or [0])
            for v in outs:
                v.creator = self
                v.rank = self.rank
        return outs[0] if len(outs) == 1 else tuple(outs)
