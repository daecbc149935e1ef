"""template_optimizer — mirror of reference Optimizers/__init__.py:10-79 for the HIP engine."""
from typing import Tuple

import numpy as np

from ..computation_library import ComputationLibrary, HipLibrary
from ..others.globals_and_utils import create_rng
from .._capi import CtkEngine, PARAMS


def logging_kwargs(kwargs: dict) -> dict:
    """the optional YAML keys of this build that every optimizer forwards to template_optimizer"""
    return {k: kwargs[k] for k in ("logging_on_device", "logging_capacity") if k in kwargs}


class DeviceLogEntry:
    """One logged tensor of one MPC step, resident in the engine's HBM log ring (ctk_log_enable).  Quacks like
    the device tensors the reference's optimizers put into `logging_values` (`.numpy()`, which
    template_controller.update_logs calls, reference Controllers/__init__.py:170-178) and like an array
    (`__array__`, `.shape`, `.copy()`); the transfer happens only when somebody looks."""
    __slots__ = ("engine", "name", "step", "shape")

    def __init__(self, engine, name: str, step: int, shape):
        self.engine, self.name, self.step, self.shape = engine, name, step, tuple(shape)

    def numpy(self) -> np.ndarray:
        return self.engine.log_read(self.name, self.step, 1)[0]

    def copy(self) -> np.ndarray:
        return self.numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    @staticmethod
    def gather(entries) -> np.ndarray:
        """Stack a list of entries along axis 0 with one transfer per run of consecutive steps
        (template_controller.get_outputs, reference Controllers/__init__.py:159-168)."""
        out, i = [], 0
        while i < len(entries):
            j = i
            while (j + 1 < len(entries) and entries[j + 1].engine is entries[i].engine
                   and entries[j + 1].name == entries[i].name and entries[j + 1].step == entries[j].step + 1):
                j += 1
            out.append(entries[i].engine.log_read(entries[i].name, entries[i].step, j - i + 1))
            i = j + 1
        return np.concatenate(out, axis=0)


class template_optimizer:
    supported_computation_libraries = (HipLibrary,)
    engine_name = None   # "mppi" | "cem" | "rpgd" | "random_action"

    def __init__(self, predictor, cost_function, control_limits: "Tuple[np.ndarray, np.ndarray]",
                 optimizer_logging: bool, seed, num_rollouts: int, mpc_horizon: int,
                 computation_library: "ComputationLibrary", rng_mode: str = "device", device: int = 0,
                 calculate_optimal_trajectory: bool = False, logging_on_device: bool = False,
                 logging_capacity: int = 4096, **kwargs) -> None:
        # reference :27-28
        if not isinstance(computation_library, self.supported_computation_libraries):
            raise ValueError(f"The optimizer {self.__class__.__name__} does not support "
                             f"{getattr(computation_library, 'lib', computation_library)}")
        self.lib = computation_library
        self.num_rollouts = num_rollouts
        self.mpc_horizon = mpc_horizon
        self.cost_function = cost_function
        self.u = 0.0
        self.predictor = predictor
        self.num_states = None
        self.num_control_inputs = None
        self.action_low, self.action_high = control_limits
        self.action_low = self.lib.to_tensor(self.action_low, self.lib.float32)
        self.action_high = self.lib.to_tensor(self.action_high, self.lib.float32)
        self.rng = create_rng(self.__class__.__name__, seed, computation_library=computation_library, mode=rng_mode)
        self.seed = getattr(self.rng, "seed", 0 if seed is None else seed)
        self.logging_values = {}
        self.optimizer_logging = optimizer_logging
        # logging_on_device: keep Q / J / trajectories / ages of every step in an HBM ring (capacity in steps) and
        # hand out DeviceLogEntry handles instead of copying ~1 MB to the host per step
        self.logging_on_device = bool(logging_on_device) and bool(optimizer_logging)
        self.logging_capacity = int(logging_capacity)
        self.calculate_optimal_trajectory = bool(calculate_optimal_trajectory)
        self.device = device
        self.engine: CtkEngine = None
        self._param_cache = {}
        self._cost_version = None
        self._sync_key = None
        self.optimal_trajectory = None
        self.rollout_trajectories = None
        self._lazy = {}   # device buffers fetched on first access after a step (u_nom: no D2H copy on the step path)

    # The reference converts u_nom to NumPy every step (optimizer_mppi.py:220, optimizer_rpgd.py:426,435); here the
    # plan stays in HBM and is copied when somebody looks at it.
    def _lazy_read(self, name):
        v = self._lazy.get(name)
        if v is None and self.engine is not None:
            v = self._lazy[name] = self.engine.read(name)
        return v

    @property
    def u_nom(self):
        return self._lazy_read("U_NOM")

    @u_nom.setter
    def u_nom(self, value):
        if value is None:
            self._lazy.pop("U_NOM", None)
        else:
            self._lazy["U_NOM"] = value

    @property
    def optimal_control_sequence(self):
        return self.u_nom

    # reference :52-63
    def configure(self, num_states: int, num_control_inputs: int, default_configure: bool = True, **kwargs) -> None:
        self.num_states = num_states
        self.num_control_inputs = num_control_inputs
        if default_configure:
            self.optimizer_reset()

    def step(self, s: np.ndarray, time=None):
        raise NotImplementedError("Implement this function in a subclass.")

    def optimizer_reset(self):
        raise NotImplementedError("Implement this function in a subclass.")

    @property
    def optimizer_name(self):
        name = self.__class__.__name__
        if name != "template_optimizer":
            return name.replace("optimizer_", "").replace("_", "-").lower()
        raise AttributeError()

    # ---- engine plumbing shared by the *_hip optimizers --------------------------------------
    def _limits(self):
        lo = np.asarray(self.action_low, np.float32).reshape(-1)
        hi = np.asarray(self.action_high, np.float32).reshape(-1)
        if lo.size != 1 or hi.size != 1:
            raise NotImplementedError("only num_control_inputs == 1 is built")
        return float(lo[0]), float(hi[0])

    def _build_engine(self, dt, predictor_specification, **engine_kwargs):
        if self.num_states != 4 or self.num_control_inputs != 1:
            raise NotImplementedError("the HIP engine is built for num_states == 4, num_control_inputs == 1")
        if getattr(self.predictor, "kind", None) is None:
            self.predictor.configure(batch_size=self.num_rollouts, dt=dt, computation_library=self.lib,
                                     predictor_specification=predictor_specification)
        lo, hi = self._limits()
        self.engine = CtkEngine(
            self.engine_name, self.predictor.kind, num_rollouts=self.num_rollouts, mpc_horizon=self.mpc_horizon,
            dt=dt, action_low=lo, action_high=hi, seed=self.seed, device=self.device,
            intermediate_steps=getattr(self.predictor, "intermediate_steps", 1),
            materialize_trajectories=bool(self.optimizer_logging), **engine_kwargs)
        if self.predictor.kind in ("MLP", "GRU"):
            self.engine.set_predictor_weights(self.predictor.weights)
        if self.logging_on_device:
            self.engine.log_enable(self.logging_capacity)
        self._param_cache = {}
        self._cost_version = None
        self._sync_parameters(force=True)

    def _sync_parameters(self, force=False):
        """Upload changed dynamics / cost / per-step attributes (reference: variable_parameters
        updated by template_controller.update_attributes, Controllers/__init__.py:106-107; cost
        YAML hot reload, cost_function_wrapper.py:71-74).  Only between steps."""
        cf = self.cost_function
        vp = getattr(cf, "variable_parameters", None) or getattr(self.predictor, "variable_parameters", None)
        # cheap per-step check: nothing to upload unless the cost parameters were reloaded (version counter), the
        # dynamics values changed, or a per-step attribute moved
        def scalar(x):
            return None if x is None else float(np.asarray(x).reshape(-1)[0])
        key = (getattr(cf, "version", None), tuple(getattr(cf, "parameters", {}).values()),
               tuple(getattr(self.predictor, "parameters", {}).values()),
               scalar(getattr(vp, "target_position", None)), scalar(getattr(vp, "target_equilibrium", None)))
        if not force and key == self._sync_key:
            return
        self._sync_key = key
        vals = {}
        vals.update(getattr(self.predictor, "parameters", {}))
        cf = self.cost_function
        vals.update(getattr(cf, "parameters", {}))
        vp = getattr(cf, "variable_parameters", None) or getattr(self.predictor, "variable_parameters", None)
        for name in ("target_position", "target_equilibrium"):
            if vp is not None and hasattr(vp, name):
                vals[name] = float(np.asarray(getattr(vp, name)).reshape(-1)[0])
        for name, v in vals.items():
            if name in PARAMS and (force or self._param_cache.get(name) != float(v)):
                self.engine.set_param(name, float(v))
                self._param_cache[name] = float(v)

    def _draws(self, kind: str, shape):
        """None => on-device Philox; otherwise RAW host draws of `shape`."""
        if getattr(self.rng, "on_device", False):
            return None
        return (self.rng.normal if kind == "normal" else self.rng.uniform)(list(shape), dtype=self.lib.float32)

    def _prepare_state(self, s):
        s = np.asarray(s, dtype=np.float32)
        if s.ndim == 2 and s.shape[0] == 1:
            s = s[0]
        if s.shape != (4,):
            raise ValueError(f"state must have shape (4,), got {s.shape}")
        return s

    def _logged(self, name: str):
        """the tensor `name` of the step just completed: a host array, or its handle in the device log"""
        if not self.logging_on_device:
            return self.engine.read(name)
        N, H = self.num_rollouts, self.mpc_horizon
        shape = {"Q": (N, H, 1), "J": (N,), "TRAJ": (N, H + 1, 4), "AGES": (N,)}[name]
        return DeviceLogEntry(self.engine, name, self.engine.log_count() - 1, shape)

    def _fill_logging(self, s_in, u):
        # reference optimizer_mppi.py:214-218 / optimizer_cem_tf.py:104-108
        self.rollout_trajectories = self._logged("TRAJ")
        self.logging_values["Q_logged"] = self._logged("Q")
        self.logging_values["J_logged"] = self._logged("J")
        self.logging_values["rollout_trajectories_logged"] = self.rollout_trajectories
        self.logging_values["u_logged"] = u

    def _predict_optimal_trajectory(self, s, u_nom, u_prev):
        # reference optimizer_mppi.py:199-202: single-trajectory rollout of the nominal plan
        traj, _ = self.engine.rollout(s, np.asarray(u_nom, np.float32).reshape(1, self.mpc_horizon, 1), u_prev=u_prev)
        return traj
