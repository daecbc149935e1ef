"""Host-side helpers mirroring reference others/globals_and_utils.py (create_rng :86-99,
optimizer discovery by file name :103-133)."""
import glob
import logging
import os
from datetime import datetime
from importlib import import_module

import numpy as np
from numpy.random import SFC64, Generator

from ..computation_library import HipLibrary


def get_logger(name):
    return logging.getLogger(name)


log = get_logger(__name__)


class DeviceRng:
    """Marker: draw on the GPU with the engine's counter-based Philox4x32-10 generator keyed by
    `seed` (performance mode).  Nothing is generated on the host."""
    on_device = True

    def __init__(self, seed: int):
        self.seed = int(seed)


class HostRng:
    """Host generator with the interface the reference's torch_gen_like_TF exposes
    (globals_and_utils.py:61-83): normal(shape, dtype) / uniform(shape, dtype) returning RAW
    N(0,1) / U[0,1) draws; scaling happens on the device exactly as the reference scales them."""
    on_device = False

    def __init__(self, seed: int):
        self.gen = Generator(SFC64(seed=seed))    # what create_rng returns for NumPy (:93-94)

    def normal(self, shape, dtype=np.float32):
        return self.gen.standard_normal(size=tuple(shape), dtype=np.float32)

    def uniform(self, shape, dtype=np.float32):
        return self.gen.random(size=tuple(shape), dtype=np.float32)


class DeviceBufferRng:
    """Draws that are ALREADY resident in HBM (the "[N, H, action_dim] sample buffer" of BASELINE.json's north_star):
    normal()/uniform() hand out the device address of the next buffer of a caller-owned pool, which the engine reads
    with coalesced loads (CTK_LOC_DEVICE).  The caller guarantees that every buffer holds at least prod(shape) fp32
    raw draws of the right distribution and outlives the step; nothing is generated or copied here."""
    on_device = False

    def __init__(self, device_pointers, seed: int = 0):
        self.pointers = [int(p) for p in device_pointers]
        if not self.pointers:
            raise ValueError("DeviceBufferRng needs at least one device buffer")
        self.seed, self._i = int(seed), 0

    def normal(self, shape, dtype=np.float32):
        p = self.pointers[self._i]
        self._i = (self._i + 1) % len(self.pointers)
        return p

    uniform = normal


def create_rng(id: str, seed, computation_library=None, mode: str = "device"):
    """reference create_rng (:86-99): seed None -> milliseconds since the epoch."""
    if seed is None:
        log.info(f"{id}: No random seed specified. Seeding with datetime.")
        seed = int((datetime.now() - datetime(1970, 1, 1)).total_seconds() * 1000.0)
    tag = str(getattr(computation_library, "lib", "")).lower()
    if computation_library is not None and not (isinstance(computation_library, HipLibrary) or tag in ("hip", "numpy")):
        raise ValueError(f"create_rng: unsupported computation library {computation_library}")
    if mode == "device":
        return DeviceRng(seed)
    if mode == "host":
        return HostRng(seed)
    raise ValueError(f"rng mode must be 'device' or 'host', got {mode!r}")


_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find_optimizer_if_it_exists(optimizer_name: str):
    """reference :103-123, searching this package's Optimizers/ folder by FILE NAME
    (optimizer_<name>.py, class name == file stem)."""
    optimizer_name = optimizer_name.replace("-", "_")
    full = optimizer_name if optimizer_name.startswith("optimizer") else "optimizer_" + optimizer_name
    paths = glob.glob(os.path.join(_PKG_DIR, "Optimizers", full + ".py"))
    if len(paths) > 1:
        raise ValueError(f"Optimizer {full} must be in a unique location. {len(paths)} found.")
    if len(paths) == 1:
        return full, paths[0]
    return False, None


def import_optimizer_by_name(optimizer_name: str) -> type:
    full, path = find_optimizer_if_it_exists(optimizer_name)
    if full:
        mod = import_module(f"control_toolkit_amd.Optimizers.{full}")
        return getattr(mod, full)
    raise ValueError(f"Optimizer {optimizer_name} not found.")
