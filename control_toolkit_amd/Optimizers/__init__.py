"""template_optimizer — mirror of reference Optimizers/__init__.py:10-79 for the HIP engine."""
from typing import Tuple

import numpy as np

from ..computation_library import ComputationLibrary, HipLibrary
from ..others.globals_and_utils import create_rng
from .._capi import CtkEngine, ENVIRONMENTS, environment_info


def predictor_kind(predictor_specification) -> str:
    """'ODE' | 'MLP' | 'GRU' from a predictor_specification string (reference controller_mpc.py:67-73; network names follow
    the convention 'GRU-6IN-32H1-32H2-5OUT-0', Control_Toolkit_ASF_Template/config_controllers.yml:8)."""
    from ..Predictors import parse_predictor_specification
    return parse_predictor_specification(predictor_specification)[0]


def caller_side_library_ok(lib) -> bool:
    """The HIP optimizers compute in the kernels; of the computation library they only use what the CALLER side touches
    (to_tensor / to_numpy / float32).  Besides HipLibrary any NumPy-semantics library object does — in particular SI_Toolkit's
    NumpyLibrary, which is what the reference's template_controller builds for `computation_library: numpy`
    (Controllers/__init__.py:55-56), so the unmodified reference controller can drive these optimizers."""
    return isinstance(lib, HipLibrary) or str(getattr(lib, "lib", "")).lower() in ("hip", "numpy")


def logging_kwargs(kwargs: dict) -> dict:
    """the optional YAML keys of this build that every optimizer forwards to template_optimizer"""
    return {k: kwargs[k] for k in ("logging_on_device", "logging_capacity", "environment_name", "generic_kernels",
                                   "predictor_parameters", "predictor_weights_file", "predictor_intermediate_steps") if k in kwargs}


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
        if not (isinstance(computation_library, self.supported_computation_libraries) or caller_side_library_ok(computation_library)):
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
        # optional keys of this build in the optimizer's YAML entry (swallowed by **kwargs in the reference's ctor too)
        self._engine_options = {k: kwargs[k] for k in ("environment_name", "generic_kernels", "predictor_parameters",
                                                       "predictor_weights_file", "predictor_intermediate_steps") if k in kwargs}
        self.engine: CtkEngine = None
        self._param_cache = {}
        self._cost_version = None
        self._sync_key = None
        self._vp_names, self._vp_len = None, -1
        self._u_public, self._u_vec = None, None
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
        """control_limits as fp32 arrays [C] (reference Optimizers/__init__.py:42-44 keeps them as tensors of that shape)"""
        C = int(self.num_control_inputs)
        lo = np.asarray(self.action_low, np.float32).reshape(-1)
        hi = np.asarray(self.action_high, np.float32).reshape(-1)
        if lo.size not in (1, C) or hi.size not in (1, C):
            raise ValueError(f"control_limits must have {C} entries (num_control_inputs), got {lo.size} / {hi.size}")
        return np.broadcast_to(lo, (C,)).copy(), np.broadcast_to(hi, (C,)).copy()

    def _resolve_environment(self) -> str:
        """Which built environment the caller means: cost_function.environment_name is what controller_mpc hands the
        cost wrapper (reference controller_mpc.py:75-82); the predictor may name it too; the optimizer's YAML entry may
        pin it (`environment_name:`).  Names are matched case-insensitively on their stem."""
        name = (self._engine_options.get("environment_name") or getattr(self.cost_function, "environment_name", None)
                or getattr(self.predictor, "environment_name", None) or "CartPole")
        stem = str(name).replace("-", "").replace("_", "").lower()
        from .._capi import USER_ENVIRONMENTS
        for built in list(ENVIRONMENTS) + list(USER_ENVIRONMENTS):
            if stem.startswith(built.replace("_", "").lower()):
                return built
        raise NotImplementedError(f"environment {name!r} is not built into libctk_hip.so (have: {sorted(ENVIRONMENTS)}) nor registered as a user "
                                  f"environment ({sorted(USER_ENVIRONMENTS)}; control_toolkit_amd.build_env.register_environment)")

    def _resolve_predictor(self, dt, predictor_specification):
        """(kind, intermediate_steps, weights) from whatever predictor object the controller passed in.  The build's own
        PredictorWrapper carries them; for a reference-shaped one (SI_Toolkit's PredictorWrapper: configured with
        predictor_specification, exposes num_states / num_control_inputs) the kind comes from the specification string
        and weights / sub-steps from the optimizer's own YAML entry (`predictor_weights_file`, `predictor_intermediate_steps`)
        — the rollout itself lives in the kernels, so nothing else of that object is needed."""
        pred = self.predictor
        if getattr(pred, "kind", None) not in ("ODE", "MLP", "GRU"):
            if hasattr(pred, "configure") and getattr(pred, "batch_size", None) is None:
                pred.configure(batch_size=self.num_rollouts, dt=dt, computation_library=self.lib,
                               predictor_specification=predictor_specification)
        kind = getattr(pred, "kind", None)
        if kind not in ("ODE", "MLP", "GRU"):
            kind = predictor_kind(predictor_specification)
        # hidden widths: what the build's wrapper parsed, else what the specification's name states (`Dense-5IN-16H1-16H2-4OUT-0`), else 32 / 32
        from ..Predictors import parse_predictor_specification, check_network_sizes
        self._hidden_sizes = getattr(pred, "hidden_sizes", None)
        if self._hidden_sizes is None and kind != "ODE":
            self._hidden_sizes = check_network_sizes(predictor_specification, parse_predictor_specification(predictor_specification)[1],
                                                     self.num_states, self.num_control_inputs, kind)
        weights = getattr(pred, "weights", None)
        wf = self._engine_options.get("predictor_weights_file")
        if weights is None and wf:
            data = np.load(wf, allow_pickle=False)
            weights = data["weights"] if hasattr(data, "files") else data
        isteps = self._engine_options.get("predictor_intermediate_steps", getattr(pred, "intermediate_steps", 1))
        return kind, int(isteps), weights

    def _build_engine(self, dt, predictor_specification, **engine_kwargs):
        env = self._resolve_environment()
        S, C, _ = environment_info(env)
        if self.num_states != S or self.num_control_inputs != C:
            raise ValueError(f"environment {env} has num_states == {S}, num_control_inputs == {C}; the predictor reports "
                             f"{self.num_states} / {self.num_control_inputs}")
        kind, isteps, weights = self._resolve_predictor(dt, predictor_specification)
        lo, hi = self._limits()
        self.engine = CtkEngine(
            self.engine_name, kind, num_rollouts=self.num_rollouts, mpc_horizon=self.mpc_horizon,
            dt=dt, action_low=lo, action_high=hi, seed=self.seed, device=self.device, intermediate_steps=isteps,
            materialize_trajectories=bool(self.optimizer_logging), environment=env,
            generic_kernels=bool(self._engine_options.get("generic_kernels", False)),
            predictor_hidden=getattr(self, "_hidden_sizes", None) if kind != "ODE" else None, **engine_kwargs)
        if kind in ("MLP", "GRU"):
            if weights is None:
                raise ValueError(f"{kind} predictor: no weights (PredictorWrapper(weights=...) or `predictor_weights_file:` in the optimizer's YAML entry)")
            self.engine.set_predictor_weights(weights, hidden=getattr(self, "_hidden_sizes", None))
        if self.logging_on_device:
            self.engine.log_enable(self.logging_capacity)
        self._param_cache = {}
        self._cost_version = None
        self._sync_key = None
        self._sync_parameters(force=True)

    def _parameter_values(self) -> dict:
        """Every value the kernels take as a constant, from the objects the controller wired in.  Provider chain, later
        wins: predictor.parameters (dynamics) < the optimizer YAML's `predictor_parameters` < the cost function's values <
        per-step attributes of variable_parameters (template_controller.update_attributes, Controllers/__init__.py:106-107).
        Cost values: the build's wrapper exposes `.parameters`; a reference-shaped wrapper holds the concrete cost object
        at `.cost_function` (cost_function_wrapper.py:62-66), whose YAML section the reference's updater stores in `.config`
        (CostFunctionUpdater.py:63-66) and whose weights are plain attributes — both are read, by name."""
        names = self.engine.param_names
        vals = {}
        vals.update(getattr(self.predictor, "parameters", None) or {})
        vals.update(self._engine_options.get("predictor_parameters") or {})
        cf = self.cost_function
        if isinstance(getattr(cf, "parameters", None), dict):
            vals.update(cf.parameters)
        else:
            inner = getattr(cf, "cost_function", None)
            for n in names:
                v = getattr(inner, n, None)
                if isinstance(v, (int, float, np.floating, np.integer)):
                    vals[n] = float(v)
            vals.update(self._reference_cost_yaml(cf))
            cfgd = getattr(inner, "config", None)
            if isinstance(cfgd, dict):
                vals.update({k: v for k, v in cfgd.items() if isinstance(v, (int, float))})
        vp = getattr(cf, "variable_parameters", None) or getattr(self.predictor, "variable_parameters", None)
        if vp is not None:
            for n in names:
                v = getattr(vp, n, None)
                if v is not None:
                    try:
                        vals[n] = float(np.asarray(v).reshape(-1)[0])
                    except (TypeError, ValueError):
                        pass
        return {k: float(v) for k, v in vals.items() if k in names}

    def _reference_cost_yaml(self, cf) -> dict:
        """Control_Toolkit_ASF/config_cost_function.yml[<environment>][<cost function name>] — the file the reference's
        cost functions take their weights from (cost_function_wrapper.py:14; CWD-relative like there), re-read when it changes."""
        import os
        path = os.path.join("Control_Toolkit_ASF", "config_cost_function.yml")
        env, name = getattr(cf, "environment_name", None), getattr(cf, "cost_function_name", None)
        if env is None or name is None or not os.path.isfile(path):
            return {}
        st = os.stat(path)
        key = (st.st_mtime_ns, st.st_size, env, name)
        if getattr(self, "_cost_yaml_key", None) != key:
            from yaml import safe_load
            try:
                section = (safe_load(open(path, "r")) or {}).get(env, {}).get(name, {}) or {}
            except Exception:   # noqa: BLE001 — a half-written file: keep what was read last
                return getattr(self, "_cost_yaml_vals", {})
            self._cost_yaml_key = key
            self._cost_yaml_vals = {k: v for k, v in section.items() if isinstance(v, (int, float))}
        return self._cost_yaml_vals

    def _sync_parameters(self, force=False):
        """Upload changed dynamics / cost / per-step attributes (reference: variable_parameters
        updated by template_controller.update_attributes, Controllers/__init__.py:106-107; cost
        YAML hot reload, cost_function_wrapper.py:71-74).  Only between steps."""
        cf, vals = self.cost_function, None
        if hasattr(cf, "version") and isinstance(getattr(cf, "parameters", None), dict):
            # fast per-step check for the build's own wrappers: the cost parameters carry a version counter, the dynamics are a
            # small dict, and of variable_parameters only the attributes that name engine parameters matter
            vp = getattr(cf, "variable_parameters", None) or getattr(self.predictor, "variable_parameters", None)
            vpd = getattr(vp, "__dict__", {}) if vp is not None else {}
            if self._vp_names is None or len(vpd) != self._vp_len:
                self._vp_names, self._vp_len = [n for n in self.engine.param_names if n in vpd], len(vpd)
            key = (cf.version, tuple(cf.parameters.values()), tuple((getattr(self.predictor, "parameters", None) or {}).values()),
                   tuple([vpd[n].item() if hasattr(vpd[n], "item") else float(vpd[n]) for n in self._vp_names]))
        else:
            vals = self._parameter_values()
            key = tuple(vals.values())
        if not force and key == self._sync_key:
            return
        self._sync_key = key
        if vals is None:
            vals = self._parameter_values()
        for name, v in vals.items():
            if force or self._param_cache.get(name) != v:
                self.engine.set_param(name, v)
                self._param_cache[name] = v

    def _draws(self, kind: str, shape):
        """None => on-device Philox; otherwise RAW host draws of `shape`."""
        if getattr(self.rng, "on_device", False):
            return None
        return (self.rng.normal if kind == "normal" else self.rng.uniform)(list(shape), dtype=self.lib.float32)

    def _prepare_state(self, s):
        s = np.asarray(s, dtype=np.float32)
        if s.ndim == 2 and s.shape[0] == 1:
            s = s[0]
        if s.shape != (self.num_states,):
            raise ValueError(f"state must have shape ({self.num_states},), got {s.shape}")
        return s

    def _u_prev(self):
        """the previous output as [C] (self.u starts as the scalar 0.0, reference Optimizers/__init__.py:34).  Fast path: self.u
        is still the object _publish_u() put there, so the engine's own [C] array is the answer; otherwise (a caller assigned
        self.u) convert."""
        if self.u is self._u_public:
            return self._u_vec
        u = np.asarray(self.u, np.float32).reshape(-1)
        return np.broadcast_to(u, (self.num_control_inputs,)) if u.size == 1 else u

    def _publish_u(self, u_vec):
        """self.u as the reference exposes it (squeezed: a scalar for one control input, optimizer_mppi.py:212) + its [C] form"""
        self._u_vec = u_vec
        self.u = self._u_public = u_vec[0] if u_vec.size == 1 else u_vec
        return self.u

    def _logged(self, name: str):
        """the tensor `name` of the step just completed: a host array, or its handle in the device log"""
        if not self.logging_on_device:
            # RPGD logs its ages at optimizer_rpgd.py:432, BEFORE the step's keep-k gather and +1 (:456-458,:514)
            return self.engine.read("AGES_LOGGED" if name == "AGES" else name)
        N, H = self.num_rollouts, self.mpc_horizon
        shape = {"Q": (N, H, self.num_control_inputs), "J": (N,), "TRAJ": (N, H + 1, self.num_states), "AGES": (N,)}[name]
        return DeviceLogEntry(self.engine, name, self.engine.log_count() - 1, shape)

    def _fill_logging(self, s_in, u):
        # reference optimizer_mppi.py:214-218 / optimizer_cem_tf.py:104-108
        self.rollout_trajectories = self._logged("TRAJ")
        self.logging_values["Q_logged"] = self._logged("Q")
        self.logging_values["J_logged"] = self._logged("J")
        self.logging_values["rollout_trajectories_logged"] = self.rollout_trajectories
        self.logging_values["u_logged"] = u

    def _predict_optimal_trajectory(self, s, u_nom, u_prev, want_summed_stage_cost: bool = False):
        # reference optimizer_mppi.py:199-202: single-trajectory rollout of the nominal plan
        plan = np.asarray(u_nom, np.float32).reshape(1, self.mpc_horizon, self.num_control_inputs)
        traj, _ = self.engine.rollout(s, plan, u_prev=u_prev)
        if not want_summed_stage_cost:
            return traj
        # optimizer_rpgd.py:382-386: cost_function.get_summed_stage_cost(optimal_trajectory, u_nom, u) = the sum of the H stage costs
        # WITHOUT the terminal cost (Cost_Functions/__init__.py:71-72).  The rollout kernels return J = (sum of stage costs + terminal)
        # / (H + 1); every built environment's terminal cost is terminal_weight * (state terms), so the same plan rolled out once more
        # with terminal_weight = 0 gives the sum of the stage costs alone — on the device, through the same kernel
        tw = self.engine.get_param("terminal_weight")
        self.engine.set_param("terminal_weight", 0.0)
        try:
            _, j0 = self.engine.rollout(s, plan, u_prev=u_prev, want_traj=False)
        finally:
            self.engine.set_param("terminal_weight", tw)
        return traj, (j0 * np.float32(self.mpc_horizon + 1)).astype(np.float32)
