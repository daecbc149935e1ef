"""Build-authored, test-only stand-in for the `tensorflow` names the reference's TF-only optimizers touch
(`Optimizers/optimizer_cem_tf.py`, `optimizer_random_action_tf.py`, `optimizer_cem_naive_grad_tf.py`):

    argsort, clip_by_value, clip_by_norm, concat, constant, convert_to_tensor, ensure_shape, float32, gather,
    math.reduce_std, math.reduce_mean, multiply, ones, reduce_mean, squeeze, tile, zeros, GradientTape,
    random.Generator.from_seed(...).normal / .uniform, Tensor, Variable (annotation only)

Tensors are torch-CPU fp32 tensors.  It exists ONLY so that `tests/golden/make_golden.py` can execute those
UNMODIFIED reference modules in the build container and record their inputs/outputs; nothing here is
TensorFlow source and nothing here travels into the product.  What such a recording pins is the reference's
own statement sequence (order of sampling, clipping, rollout, sort, refit, shift, which row becomes `u`);
the meaning of each `tf.*` primitive is this file's, chosen to follow TensorFlow's documented definitions:

  * `argsort`: ascending and STABLE (TensorFlow's implementation is a stable sort whatever `stable=` says),
    i.e. the total order (cost, index);
  * `math.reduce_std`: POPULATION standard deviation, sqrt(mean((x - mean(x))^2)) (no Bessel correction);
  * `clip_by_norm(t, c, axes)`: t * c / max(||t||_2 over axes, c);
  * `random.Generator.normal/uniform`: the arithmetic of the reference's own torch generator
    (`others/globals_and_utils.py:61-83`): N(0,1)*stddev + mean; U[0,1)*(maxval - minval) + minval.  The bit
    stream is torch's (TensorFlow's Philox stream is not reproducible here and the tests feed the kernels the
    recorded draws, so only this arithmetic matters).
"""
import numpy as _np
import torch as _torch

float32 = _torch.float32
int32 = _torch.int32
int64 = _torch.int64
Tensor = _torch.Tensor
Variable = _torch.Tensor


def _t(x, dtype=None):
    if isinstance(x, _torch.Tensor):
        return x if dtype is None else x.to(dtype)
    return _torch.as_tensor(_np.asarray(x), dtype=dtype)


def constant(value, dtype=None, shape=None):
    if dtype is None and not isinstance(value, _torch.Tensor):
        a = _np.asarray(value)
        if a.dtype.kind in "iu":
            # integer constants are only used as `reps` of np.tile (optimizer_cem_tf.py:87): hand numpy an array it can iterate
            return a if shape is None else a.reshape(shape)
        dtype = float32
    out = _t(value, dtype)
    return out if shape is None else out.reshape(tuple(shape))


def convert_to_tensor(value, dtype=None):
    return _t(value, dtype).clone()


def zeros(shape, dtype=float32):
    return _torch.zeros(tuple(_np.atleast_1d(shape).tolist()), dtype=dtype)


def ones(shape, dtype=float32):
    return _torch.ones(tuple(_np.atleast_1d(shape).tolist()), dtype=dtype)


def tile(x, multiples):
    return _torch.tile(x, tuple(int(m) for m in multiples))


def multiply(a, b):
    return a * b


def clip_by_value(x, clip_value_min, clip_value_max):
    return _torch.clamp(x, min=clip_value_min, max=clip_value_max)


def clip_by_norm(t, clip_norm, axes=None):
    dims = tuple(axes) if axes is not None else tuple(range(t.ndim))
    nrm = _torch.sqrt(_torch.sum(t * t, dim=dims, keepdim=True))
    c = _t(clip_norm, t.dtype)
    return t * c / _torch.maximum(nrm, c)


def concat(values, axis):
    return _torch.cat([_t(v, float32) for v in values], dim=axis)


def squeeze(x):
    return _torch.squeeze(x)


def ensure_shape(x, shape):
    assert tuple(x.shape) == tuple(int(s) for s in shape), (tuple(x.shape), tuple(shape))
    return x


def argsort(values, axis=-1, direction="ASCENDING", stable=False):
    assert direction == "ASCENDING"
    return _torch.argsort(values, dim=axis, stable=True)


def gather(params, indices, axis=0):
    return _torch.index_select(params, axis, indices)


def reduce_mean(x, axis=None, keepdims=False):
    return _torch.mean(x, dim=axis, keepdim=keepdims)


class _Math:
    reduce_mean = staticmethod(reduce_mean)

    @staticmethod
    def reduce_std(x, axis=None, keepdims=False):
        mean = _torch.mean(x, dim=axis, keepdim=True)
        var = _torch.mean((x - mean) * (x - mean), dim=axis, keepdim=keepdims)
        return _torch.sqrt(var)


math = _Math()


class GradientTape:
    """`with tf.GradientTape(watch_accessed_variables=False) as tape: tape.watch(Q); ...; tape.gradient(J, Q)`:
    torch autograd of sum(J) with respect to the watched tensor; the tensor stops recording once the gradient is taken,
    as everything outside the `with` block does under TensorFlow."""

    def __init__(self, watch_accessed_variables=True, persistent=False):
        self._watched = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def watch(self, x):
        x.requires_grad_(True)
        self._watched.append(x)

    def gradient(self, target, source):
        g, = _torch.autograd.grad(target.sum(), source)
        for w in self._watched:
            w.requires_grad_(False)
        return g


class _Generator:
    def __init__(self, seed):
        self.rng = _torch.Generator().manual_seed(int(seed))

    @classmethod
    def from_seed(cls, seed):
        return cls(seed)

    def normal(self, shape, mean=0.0, stddev=1.0, dtype=float32):
        shape = tuple(int(s) for s in shape)
        return _torch.normal(mean=0.0, std=1.0, size=shape, generator=self.rng, dtype=dtype) * stddev + mean

    def uniform(self, shape, minval=0.0, maxval=1.0, dtype=float32):
        shape = tuple(int(s) for s in shape)
        return _torch.rand(*shape, generator=self.rng, dtype=dtype) * (maxval - minval) + minval


class _Random:
    Generator = _Generator


random = _Random()
