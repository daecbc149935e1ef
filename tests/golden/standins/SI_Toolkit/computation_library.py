"""Thin torch-CPU adapter exposing the `lib.*` members the reference optimizers call
(list: SURVEY.md 8b).  Semantics follow the TensorFlow names the reference was written against."""
import math
import numpy as np
import torch

TensorType = torch.Tensor
VariableType = torch.Tensor


class ComputationLibrary:
    lib = None


class NumpyLibrary(ComputationLibrary):
    """caller-side members only (Controllers/__init__.py:72-82, Optimizers/__init__.py:42-44, controller_mpc.py:93-96):
    enough for the reference's controller_mpc to drive an optimizer that computes elsewhere (tests/ref_plugin_probe.py)"""
    lib = "Numpy"
    float32 = np.float32

    @staticmethod
    def set_device(device_name):
        def deco(fn):
            return fn
        return deco

    @staticmethod
    def to_tensor(x, dtype):
        return np.asarray(x, dtype=dtype)

    @staticmethod
    def to_variable(x, dtype):
        return np.array(x, dtype=dtype)

    @staticmethod
    def to_numpy(x):
        return np.asarray(x)


class _TorchBacked:
    """the `lib.*` members, on torch-CPU tensors; shared by the two tensor libraries below (which must stay unrelated
    classes: the reference branches on `isinstance(self.lib, TensorFlowLibrary)`, optimizer_rpgd.py:35,52,86)"""
    float32 = torch.float32
    int64 = torch.int64
    newaxis = None

    @staticmethod
    def set_device(device_name):
        def deco(fn):
            return fn
        return deco

    @staticmethod
    def to_tensor(x, dtype):
        if isinstance(x, torch.Tensor):
            return x.to(dtype)
        return torch.as_tensor(np.asarray(x), dtype=dtype)

    constant = to_tensor

    @staticmethod
    def to_variable(x, dtype):
        return _TorchBacked.to_tensor(x, dtype).clone()

    @staticmethod
    def to_numpy(x):
        return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)

    @staticmethod
    def assign(v, x):
        # tf.Variable.assign semantics: earlier slices of `v` are independent tensors.  A torch
        # in-place copy_ would alias `u_nom = Q_tf[None, best_idx[0]]` (a view, optimizer_rpgd.py:426)
        # with the warm-started population written at :515; rebinding the storage keeps the
        # TF (and algorithmically intended) behaviour: u = first input of the best plan.
        v.data = x.detach().clone()

    @staticmethod
    def ndim(x): return x.ndim
    @staticmethod
    def tile(x, reps): return torch.tile(x, tuple(int(r) for r in reps))
    @staticmethod
    def concat(xs, axis): return torch.cat(list(xs), dim=axis)
    @staticmethod
    def clip(x, lo, hi): return torch.clamp(x, min=lo, max=hi)
    @staticmethod
    def sum(x, axis): return torch.sum(x, dim=axis)
    @staticmethod
    def mean(x, axis): return torch.mean(x, dim=axis)
    @staticmethod
    def reduce_min(x, axis): return torch.amin(x, dim=axis)
    @staticmethod
    def exp(x): return torch.exp(x)
    @staticmethod
    def abs(x): return torch.abs(x)
    @staticmethod
    def ones(shape): return torch.ones(tuple(shape), dtype=torch.float32)
    @staticmethod
    def zeros(shape): return torch.zeros(tuple(shape), dtype=torch.float32)
    @staticmethod
    def zeros_like(x): return torch.zeros_like(x)
    @staticmethod
    def reshape(x, shape): return torch.reshape(x, tuple(shape))
    @staticmethod
    def squeeze(x): return torch.squeeze(x)
    @staticmethod
    def permute(x, perm): return x.permute(*perm)
    @staticmethod
    def matmul(a, b): return torch.matmul(a, b)
    @staticmethod
    def arange(n): return torch.arange(n)
    @staticmethod
    def cast(x, dtype): return x.to(dtype)
    @staticmethod
    def ceil(x): return math.ceil(x) if not isinstance(x, torch.Tensor) else torch.ceil(x)
    @staticmethod
    def floor(x): return math.floor(x) if not isinstance(x, torch.Tensor) else torch.floor(x)
    @staticmethod
    def argsort(x, axis=0): return torch.argsort(x, dim=axis, stable=True)   # total order (cost, index)
    @staticmethod
    def gather(x, idx, axis): return torch.index_select(x, axis, idx)

    @staticmethod
    def clip_by_norm(x, clip_norm, axes):
        # tf.clip_by_norm: x * clip / max(||x||, clip)
        nrm = torch.sqrt(torch.sum(x * x, dim=tuple(axes), keepdim=True))
        return x * clip_norm / torch.maximum(nrm, torch.as_tensor(clip_norm, dtype=x.dtype))


class PyTorchLibrary(_TorchBacked, ComputationLibrary):
    lib = "Pytorch"


class TensorFlowLibrary(_TorchBacked, ComputationLibrary):
    """`computation_library: tensorflow` in the golden runs of the TF-only optimizers (optimizer_cem_tf,
    optimizer_random_action_tf, optimizer_cem_naive_grad_tf): the same torch-backed members under the tag the
    reference expects; the `tf.*` names those modules call directly come from standins/tensorflow."""
    lib = "TF"


ComputationClasses = (NumpyLibrary, TensorFlowLibrary, PyTorchLibrary)
