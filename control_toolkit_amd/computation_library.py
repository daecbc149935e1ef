"""HipLibrary — the `computation_library` object that gates the *_hip optimizers.

The reference maps the YAML string `computation_library` to a library object
(Controllers/__init__.py:46-58) and every optimizer checks
`isinstance(computation_library, supported_computation_libraries)` (Optimizers/__init__.py:27-28).
Only the members the CALLER side touches are needed: to_tensor, to_variable,
to_numpy, float32, set_device and the string tag `lib`; the ~30 tensor ops the reference
optimizers call on their library are fused into the HIP kernels."""
import numpy as np


class ComputationLibrary:
    lib = None


class HipLibrary(ComputationLibrary):
    lib = "HIP"
    float32 = np.float32
    int32 = np.int32
    int64 = np.int64
    newaxis = np.newaxis

    @staticmethod
    def to_tensor(x, dtype=np.float32):
        return np.asarray(x, dtype=dtype)

    @staticmethod
    def to_variable(x, dtype=np.float32):
        return np.array(x, dtype=dtype)

    @staticmethod
    def to_numpy(x):
        return np.asarray(x)

    @staticmethod
    def set_device(device_name):
        """Decorator factory, as the reference uses it (controller_mpc.py:93-96).  The HIP device
        ordinal is fixed when the engine handle is created, so this only validates the name."""
        name = "" if device_name is None else str(device_name).lower()
        if name and not any(t in name for t in ("gpu", "cuda", "hip", "rocm", "mi355", "default")):
            raise ValueError(f"HipLibrary runs on MI355X only; device {device_name!r} is not a GPU device string")

        def decorator(fn):
            return fn
        return decorator

    @staticmethod
    def device_ordinal(device_name) -> int:
        name = "" if device_name is None else str(device_name)
        if ":" in name:
            try:
                return int(name.rsplit(":", 1)[1])
            except ValueError:
                return 0
        return 0


ComputationClasses = (HipLibrary,)
