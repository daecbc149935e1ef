"""control_toolkit_amd — MI355X-native batched-rollout engine behind Control_Toolkit's
sampling-based MPC optimizers (MPPI / CEM / RPGD / random-action).

The compute lives in `libctk_hip.so` (hand-written HIP for gfx950, C ABI in include/ctk_hip.h);
this package is the host-side mirror of the reference's plugin interface
(`template_optimizer`, `template_controller`, `controller_mpc`) so that
`optimizer: mppi-hip` selects it with no edits to the caller.  There is no CPU fallback.
"""
from ._capi import CtkEngine, CtkError, library_path, load_library  # noqa: F401
from .computation_library import HipLibrary  # noqa: F401

__all__ = ["CtkEngine", "CtkError", "HipLibrary", "library_path", "load_library"]
