"""ctypes binding of include/ctk_hip.h (the stub a reference maintainer would add; the reference's
own precedent for a ctypes boundary is Controllers/controller_C.py:261-274).

Fails loudly: if libctk_hip.so is missing or cannot be loaded, importing the engine raises; if
no gfx950 device is usable, `CtkEngine(...)` raises.  Nothing here computes on the CPU."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libctk_hip.so"
_lib = None

OPTIMIZERS = {"mppi": 0, "cem": 1, "rpgd": 2, "random_action": 3, "gradient": 4, "cem_naive_grad": 5,
              "cem_grad_bharadhwaj": 6}
PREDICTORS = {"ODE": 0, "MLP": 1, "GRU": 2}
ENVIRONMENTS = {"CartPole": 0, "Quad2D": 1, "Hover": 2}          # include/ctk_hip.h: enum ctk_environment
MAX_STATES, MAX_INPUTS = 8, 4
# CartPole's parameter names in id order (enum ctk_param); `environment_params(name)` asks the library for any environment's
PARAMS = ("g", "m_cart", "m_pole", "L", "u_max", "M_fric", "J_fric", "target_position", "target_equilibrium",
          "dd_weight", "ep_weight", "ekp_weight", "cc_weight", "ccrc_weight", "R", "x_scale", "terminal_weight")
BUFFERS = {"Q": 0, "J": 1, "TRAJ": 2, "U_NOM": 3, "STD": 4, "ADAM_M": 5, "ADAM_V": 6, "AGES": 7, "BEST_IDX": 8, "PLAN": 9, "AGES_LOGGED": 10}
LOC_NONE, LOC_HOST, LOC_DEVICE = 0, 1, 2
MLP_NUM_WEIGHTS = 1380


class CtkError(RuntimeError):
    pass


class CtkConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("optimizer", C.c_int32), ("predictor", C.c_int32), ("device", C.c_int32),
        ("num_rollouts", C.c_int32), ("mpc_horizon", C.c_int32), ("num_states", C.c_int32),
        ("num_control_inputs", C.c_int32), ("period_interpolation_inducing_points", C.c_int32),
        ("intermediate_steps", C.c_int32), ("materialize_trajectories", C.c_int32),
        ("global_rollout_offset", C.c_int32), ("seed", C.c_uint64), ("dt", C.c_float),
        ("environment", C.c_int32), ("generic_kernels", C.c_int32),
        ("action_low", C.c_float * MAX_INPUTS), ("action_high", C.c_float * MAX_INPUTS),
        ("cc_weight", C.c_float), ("R", C.c_float), ("LBD", C.c_float), ("NU", C.c_float), ("SQRTRHOINV", C.c_float),
        ("cem_outer_it", C.c_int32), ("cem_best_k", C.c_int32), ("warmup", C.c_int32), ("warmup_iterations", C.c_int32),
        ("cem_initial_action_stdev", C.c_float), ("cem_stdev_min", C.c_float),
        ("outer_its", C.c_int32), ("resamp_per", C.c_int32), ("shift_previous", C.c_int32), ("opt_keep_k", C.c_int32),
        ("sampling_distribution", C.c_int32), ("sample_whole_control_space", C.c_int32),
        ("sample_stdev", C.c_float), ("sample_mean", C.c_float), ("sample_min", C.c_float), ("sample_max", C.c_float),
        ("learning_rate", C.c_float), ("gradmax_clip", C.c_float), ("adam_beta_1", C.c_float),
        ("adam_beta_2", C.c_float), ("adam_epsilon", C.c_float), ("adam_rule", C.c_int32),
        ("predictor_hidden1", C.c_int32), ("predictor_hidden2", C.c_int32),
    ]


def library_path() -> str:
    """The in-tree product library.  CTK_HIP_LIBRARY points a diagnostic run (tools/rpgd_split.sh: timing variants whose
    results are meaningless) at another build WITHOUT touching the product file; it is announced on stderr so that a test or
    bench line can never be attributed to the product library by mistake."""
    override = os.environ.get("CTK_HIP_LIBRARY")
    if override:
        import sys
        print(f"[ctk] CTK_HIP_LIBRARY override: loading {override} instead of the product library", file=sys.stderr)
        return override
    return os.path.join(_HERE, _LIB_NAME)


# every symbol include/ctk_hip.h declares: name -> (restype, argtypes)
_FP = C.POINTER(C.c_float)
_H = C.c_void_p
SYMBOLS = {
    "ctk_abi_version": (C.c_int, []),
    "ctk_create": (C.c_int, [C.POINTER(CtkConfig), C.POINTER(_H)]),
    "ctk_destroy": (None, [_H]),
    "ctk_reset": (C.c_int, [_H, C.c_void_p, C.c_int]),
    "ctk_last_error": (C.c_char_p, [_H]),
    "ctk_set_stream": (C.c_int, [_H, C.c_void_p]),
    "ctk_get_stream": (C.c_void_p, [_H]),
    "ctk_set_param": (C.c_int, [_H, C.c_int, C.c_float]),
    "ctk_get_param": (C.c_int, [_H, C.c_int, _FP]),
    "ctk_env_info": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ctk_param_name": (C.c_char_p, [C.c_int, C.c_int]),
    "ctk_environment_name": (C.c_char_p, [C.c_int]),
    "ctk_param_default": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "ctk_predictor_weight_count": (C.c_size_t, [_H]),
    "ctk_set_predictor_weights": (C.c_int, [_H, C.c_void_p, C.c_size_t]),
    "ctk_predictor_weight_count_shaped": (C.c_size_t, [_H, C.c_int, C.c_int]),
    "ctk_set_predictor_weights_shaped": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.c_int, C.c_int]),
    "ctk_predictor_hidden_size": (C.c_size_t, [_H]),
    "ctk_predictor_update": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "ctk_predictor_get_hidden": (C.c_int, [_H, C.c_void_p, C.c_size_t]),
    "ctk_predictor_set_hidden": (C.c_int, [_H, C.c_void_p, C.c_size_t]),
    "ctk_step": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ctk_samples_needed": (C.c_size_t, [_H]),
    "ctk_rng_get_position": (C.c_int, [_H, C.POINTER(C.c_uint32)]),
    "ctk_rng_set_position": (C.c_int, [_H, C.c_uint32]),
    "ctk_rollout": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ctk_mppi_partial_size": (C.c_size_t, [_H]),
    "ctk_mppi_step_begin": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ctk_mppi_step_end": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p]),
    "ctk_shard_candidates_size": (C.c_size_t, [_H]),
    "ctk_shard_iterations": (C.c_int, [_H]),
    "ctk_shard_iter_begin": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ctk_shard_iter_end": (C.c_int, [_H, C.c_void_p, C.c_int]),
    "ctk_shard_finish": (C.c_int, [_H, C.c_void_p]),
    "ctk_rpgd_keepers_size": (C.c_size_t, [_H]),
    "ctk_rpgd_fresh_rows": (C.c_size_t, [_H, C.c_int]),
    "ctk_rpgd_step_begin": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ctk_rpgd_step_end": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "ctk_read": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ctk_state_size": (C.c_size_t, [_H]),
    "ctk_get_state": (C.c_int, [_H, C.c_void_p, C.c_size_t]),
    "ctk_set_state": (C.c_int, [_H, C.c_void_p, C.c_size_t]),
    "ctk_profile_enable": (C.c_int, [_H, C.c_int]),
    "ctk_profile_read": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ctk_dominant_kernel": (C.c_char_p, [_H]),
    "ctk_p2p_alloc": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p]),
    "ctk_p2p_connect": (C.c_int, [_H, C.c_void_p]),
    "ctk_p2p_step": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ctk_p2p_close": (C.c_int, [_H]),
    "ctk_log_enable": (C.c_int, [_H, C.c_size_t]),
    "ctk_log_count": (C.c_size_t, [_H]),
    "ctk_log_read": (C.c_int, [_H, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ctk_resident_enable": (C.c_int, [_H, C.c_int, C.c_double]),
    "ctk_resident_stop": (C.c_int, [_H]),
    "ctk_resident_stats": (C.c_int, [_H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}


# user environments (include/ctk_user_env.h): name -> path of the library that was compiled with that model (control_toolkit_amd/build_env.py:
# register_environment); their environment id inside that library is CTK_ENV_USER
USER_ENVIRONMENTS = {}
ENV_USER = 3
_user_libs = {}


def load_library(path: str = None):
    """Load libctk_hip.so (built in-tree by control_toolkit_amd/csrc/Makefile).  Raises CtkError
    if it is missing — there is deliberately no other implementation to fall back to.
    path: a library compiled with a user environment (build_env.py); every such library carries the whole engine."""
    global _lib
    if path is not None:
        if path not in _user_libs:
            _user_libs[path] = _bind_library(path)
        return _user_libs[path]
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one
    # process leave the second one without devices ("No HIP GPUs are available").  Import torch
    # first so that libctk_hip.so binds to the runtime already loaded (torch is only plumbing
    # here: device memory for collectives, streams, torch.distributed).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    _lib = _bind_library(library_path())
    return _lib


def _bind_library(path: str):
    try:
        import torch  # noqa: F401  (see load_library: one HIP runtime per process)
    except ImportError:
        pass
    if not os.path.exists(path):
        raise CtkError(f"{path} not found: build it with `make -C control_toolkit_amd/csrc` "
                       f"(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise CtkError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.ctk_abi_version() != 6:
        raise CtkError(f"{path}: ABI version mismatch")
    return lib


def environment_library(name: str):
    """(library, environment id) that implement environment `name`: the product library for the built ones, the library compiled
    with the model for a registered user environment"""
    if name in ENVIRONMENTS:
        return load_library(), ENVIRONMENTS[name]
    if name in USER_ENVIRONMENTS:
        return load_library(USER_ENVIRONMENTS[name]), ENV_USER
    raise NotImplementedError(f"environment {name!r} is not built (have: {sorted(ENVIRONMENTS)} + registered user environments "
                              f"{sorted(USER_ENVIRONMENTS)}; control_toolkit_amd.build_env.register_environment compiles a model header)")


def environment_info(name: str):
    """(S, C, parameter names in id order) of an environment, as the library defines them (ctk_env_info / ctk_param_name)."""
    lib, eid = environment_library(name)
    S, Cn, n = C.c_int(), C.c_int(), C.c_int()
    if lib.ctk_env_info(eid, C.byref(S), C.byref(Cn), C.byref(n)) != 0:
        raise CtkError(f"ctk_env_info({name}) failed")
    return S.value, Cn.value, tuple(lib.ctk_param_name(eid, i).decode() for i in range(n.value))


def environment_defaults(name: str) -> dict:
    """parameter name -> the value a new handle starts with (ctk_param_default)"""
    lib, eid = environment_library(name)
    _, _, names = environment_info(name)
    out, v = {}, C.c_float()
    for i, n in enumerate(names):
        if lib.ctk_param_default(eid, i, C.byref(v)) != 0:
            raise CtkError(f"ctk_param_default({name}, {i}) failed")
        out[n] = float(v.value)
    return out


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None) -> np.ndarray:
    out = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None and tuple(out.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(out.shape)}")
    return out


class CtkEngine:
    """Owns one ctk_handle (one optimizer instance on one GPU)."""

    def __init__(self, optimizer: str, predictor: str, *, num_rollouts: int, mpc_horizon: int, dt: float,
                 action_low: float = -1.0, action_high: float = 1.0, period_interpolation_inducing_points: int = 1,
                 seed: int = 0, device: int = 0, intermediate_steps: int = 1, materialize_trajectories: bool = False,
                 global_rollout_offset: int = 0, num_states: int = None, num_control_inputs: int = None,
                 environment: str = "CartPole", generic_kernels: bool = False, predictor_hidden=None, **kw):
        """action_low / action_high: scalars (every input) or one value per control input.  environment: the plant +
        cost the kernels implement ("CartPole", "Quad2D", "Hover"); num_states / num_control_inputs default to its dimensions and
        are checked against them.  generic_kernels: run the environment-agnostic template kernels even where a hand-tuned
        one exists."""
        lib, env_id = environment_library(environment)
        S, Cn, self.param_names = environment_info(environment)
        self.environment, self.S, self.C = environment, S, Cn
        num_states = S if num_states is None else num_states
        num_control_inputs = Cn if num_control_inputs is None else num_control_inputs
        if optimizer not in OPTIMIZERS:
            raise ValueError(f"unknown optimizer {optimizer!r}")
        if predictor not in PREDICTORS:
            raise NotImplementedError(f"predictor_specification {predictor!r} is not built (have: {list(PREDICTORS)})")
        cfg = CtkConfig()
        cfg.struct_size = C.sizeof(CtkConfig)
        cfg.optimizer, cfg.predictor, cfg.device = OPTIMIZERS[optimizer], PREDICTORS[predictor], device
        cfg.num_rollouts, cfg.mpc_horizon = int(num_rollouts), int(mpc_horizon)
        cfg.num_states, cfg.num_control_inputs = int(num_states), int(num_control_inputs)
        cfg.period_interpolation_inducing_points = int(period_interpolation_inducing_points)
        cfg.intermediate_steps = int(intermediate_steps)
        cfg.materialize_trajectories = int(bool(materialize_trajectories))
        cfg.global_rollout_offset = int(global_rollout_offset)
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.dt = float(dt)
        cfg.environment, cfg.generic_kernels = env_id, int(bool(generic_kernels))
        # hidden widths of a network predictor (the <h1>H1-<h2>H2 of the reference's network names): widths above 32 (MLP, up to 64) build
        # the handle on the 64-unit kernels; narrower networks are embedded exactly when their weights are set
        self.predictor_hidden = None if predictor_hidden is None else (int(predictor_hidden[0]), int(predictor_hidden[1]))
        self.native_hidden = (64, 64) if (self.predictor_hidden and max(self.predictor_hidden) > 32) else (32, 32)
        lo = np.broadcast_to(np.asarray(action_low, np.float32).reshape(-1), (Cn,)) if np.size(action_low) in (1, Cn) else None
        hi = np.broadcast_to(np.asarray(action_high, np.float32).reshape(-1), (Cn,)) if np.size(action_high) in (1, Cn) else None
        if lo is None or hi is None:
            raise ValueError(f"control limits must be scalars or have {Cn} entries (num_control_inputs of {environment})")
        for c in range(Cn):
            cfg.action_low[c], cfg.action_high[c] = float(lo[c]), float(hi[c])
        # defaults keep unrelated optimizers' fields valid
        defaults = dict(cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0, SQRTRHOINV=0.03, cem_outer_it=1, cem_best_k=1,
                        warmup=0, warmup_iterations=0, cem_initial_action_stdev=0.5, cem_stdev_min=0.01,
                        outer_its=1, resamp_per=1, shift_previous=1, opt_keep_k=1, sampling_distribution=0,
                        sample_whole_control_space=0, sample_stdev=0.5, sample_mean=0.0, sample_min=-1.0, sample_max=1.0, learning_rate=0.05,
                        gradmax_clip=5.0, adam_beta_1=0.9, adam_beta_2=0.999, adam_epsilon=1e-8, adam_rule=0, predictor_hidden1=0, predictor_hidden2=0)
        unknown = set(kw) - set(defaults)
        if unknown:
            raise TypeError(f"unknown engine arguments: {sorted(unknown)}")
        defaults.update(kw)
        if self.predictor_hidden is not None:
            defaults.update(predictor_hidden1=self.predictor_hidden[0], predictor_hidden2=self.predictor_hidden[1])
        for k, v in defaults.items():
            setattr(cfg, k, type(getattr(cfg, k))(v))
        self._lib, self.cfg = lib, cfg
        self.optimizer, self.predictor = optimizer, predictor
        self.N, self.H = int(num_rollouts), int(mpc_horizon)
        self._h = _H()
        rc = lib.ctk_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            msg = lib.ctk_last_error(None).decode()
            self._h = _H()
            raise (ValueError if rc == 1 else NotImplementedError if rc == 2 else CtkError)(msg)
        # preallocated argument buffers: the per-step call path does no allocation and no ndarray->ctypes casts
        self._u = np.zeros(Cn, np.float32)
        self._s = np.zeros(S, np.float32)
        self._up = np.zeros(Cn, np.float32)
        self._u_p, self._s_p, self._up_p = self._u.ctypes.data, self._s.ctypes.data, self._up.ctypes.data
        self._step_fn = lib.ctk_step

    # ---- plumbing ------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            msg = self._lib.ctk_last_error(self._h).decode()
            raise (ValueError if rc == 1 else NotImplementedError if rc == 2 else CtkError)(f"[ctk {rc}] {msg}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.ctk_destroy(self._h)
            self._h = _H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- API -----------------------------------------------------------------------------------
    def reset(self, draws=None, loc: int = LOC_NONE):
        if draws is not None and loc == LOC_NONE:
            loc = LOC_HOST
        d = _f32(draws) if (draws is not None and loc == LOC_HOST) else None
        self._check(self._lib.ctk_reset(self._h, _ptr(d) if d is not None else (draws if loc == LOC_DEVICE else None), loc))

    def get_stream(self) -> int:
        """the HIP stream the handle issues on right now (after resident_enable: a high-priority stream of its own)"""
        return int(self._lib.ctk_get_stream(self._h) or 0)

    def set_stream(self, stream_ptr: int):
        self._check(self._lib.ctk_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_param(self, name: str, value: float):
        self._check(self._lib.ctk_set_param(self._h, self.param_names.index(name), float(value)))

    def get_param(self, name: str) -> float:
        v = C.c_float()
        self._check(self._lib.ctk_get_param(self._h, self.param_names.index(name), C.byref(v)))
        return v.value

    def predictor_weight_count(self, hidden=None) -> int:
        if hidden is None:
            return int(self._lib.ctk_predictor_weight_count(self._h))
        return int(self._lib.ctk_predictor_weight_count_shaped(self._h, int(hidden[0]), int(hidden[1])))

    def set_predictor_weights(self, w, hidden=None):
        """hidden = (h1, h2): the network's hidden widths (the <h1>H1-<h2>H2 of the reference's network names); None = what the engine was
        created for (predictor_hidden, default 32 / 32).  Widths below the handle's own (32, or 64 for an engine created with
        predictor_hidden above 32) are embedded exactly; wider ones raise NotImplementedError with the sizes."""
        w = _f32(w).ravel()
        if hidden is None:
            hidden = self.predictor_hidden or self.native_hidden
        hidden = (int(hidden[0]), int(hidden[1]))
        if hidden == self.native_hidden:
            self._check(self._lib.ctk_set_predictor_weights(self._h, _ptr(w), w.size))
        else:
            self._check(self._lib.ctk_set_predictor_weights_shaped(self._h, _ptr(w), w.size, hidden[0], hidden[1]))
        self.hidden_sizes = hidden

    # recurrent predictor state (GRU): predictor.update(s, Q0), optimizer_mppi.py:195-197
    def predictor_hidden_size(self) -> int:
        return int(self._lib.ctk_predictor_hidden_size(self._h))

    def predictor_update(self, s, u=None):
        s = _f32(s).ravel()
        u = None if u is None else _f32(u).ravel()
        self._check(self._lib.ctk_predictor_update(self._h, _ptr(s), _ptr(u)))

    def predictor_get_hidden(self) -> np.ndarray:
        out = np.empty(self.predictor_hidden_size(), np.float32)
        self._check(self._lib.ctk_predictor_get_hidden(self._h, _ptr(out), out.size))
        return out.reshape(2, -1)

    def predictor_set_hidden(self, hidden=None):
        hid = None if hidden is None else _f32(hidden).ravel()
        self._check(self._lib.ctk_predictor_set_hidden(self._h, _ptr(hid), 0 if hid is None else hid.size))

    def samples_needed(self) -> int:
        return int(self._lib.ctk_samples_needed(self._h))

    def rng_position(self) -> int:
        v = C.c_uint32()
        self._check(self._lib.ctk_rng_get_position(self._h, C.byref(v)))
        return int(v.value)

    def set_rng_position(self, call: int):
        self._check(self._lib.ctk_rng_set_position(self._h, int(call) & 0xFFFFFFFF))

    def samples_needed_reset(self) -> int:
        """RPGD: raw draws optimizer_reset consumes (N * P * C)."""
        return self.N * int(self._lib.ctk_mppi_partial_size(self._h) - 2)

    def inducing_points(self) -> int:
        """P of the interpolator (others/Interpolator.py:79-84) as the engine uses it"""
        return int(self._lib.ctk_mppi_partial_size(self._h) - 2) // self.C

    def step(self, s, samples=None, loc: int = None, u_prev=None) -> np.ndarray:
        """samples: None (device Philox), a host ndarray (parity mode) or an int device pointer."""
        try:
            self._s[:] = np.asarray(s).reshape(-1)
        except ValueError:
            raise ValueError(f"state must have {self.S} entries") from None
        up_p = None
        if u_prev is not None:
            self._up[:] = np.asarray(u_prev).reshape(-1)[:self.C]
            up_p = self._up_p
        if samples is None:
            sp, loc = None, LOC_NONE
        elif type(samples) is int:
            sp, loc = samples, LOC_DEVICE
        else:
            arr = _f32(samples)
            need = self.samples_needed()
            if arr.size != need:
                raise ValueError(f"step consumes {need} draws, got {arr.size}")
            sp, loc = arr.ctypes.data, LOC_HOST
        rc = self._step_fn(self._h, self._s_p, up_p, sp, loc, self._u_p)
        if rc:
            self._check(rc)
        return self._u.copy()

    def rollout(self, s, Q, u_prev=0.0, want_traj=True):
        Q = _f32(Q)
        n = Q.shape[0]
        Q = np.ascontiguousarray(Q.reshape(n, self.H, self.C))
        s = _f32(s).reshape(-1)
        if s.size != self.S:
            raise ValueError(f"state must have {self.S} entries")
        up = np.ascontiguousarray(np.broadcast_to(_f32(u_prev).reshape(-1), (self.C,)))
        traj = np.empty((n, self.H + 1, self.S), np.float32) if want_traj else None
        J = np.empty((n,), np.float32)
        self._check(self._lib.ctk_rollout(self._h, _ptr(s), _ptr(up), _ptr(Q), n, _ptr(traj), _ptr(J)))
        return traj, J

    def mppi_partial_size(self) -> int:
        return int(self._lib.ctk_mppi_partial_size(self._h))

    def mppi_step_begin(self, s, partial_dev_ptr: int, samples=None, u_prev=None):
        s = _f32(s).reshape(-1)
        up = None if u_prev is None else _f32(u_prev).reshape(-1)[:self.C].copy()
        if samples is None:
            sp, loc = None, LOC_NONE
        elif isinstance(samples, int):
            sp, loc = C.c_void_p(samples), LOC_DEVICE
        else:
            arr = _f32(samples); sp, loc = _ptr(arr), LOC_HOST
        self._check(self._lib.ctk_mppi_step_begin(self._h, _ptr(s), _ptr(up), sp, loc, C.c_void_p(partial_dev_ptr)))

    def mppi_step_end(self, parts_dev_ptr: int, n_parts: int) -> np.ndarray:
        self._check(self._lib.ctk_mppi_step_end(self._h, C.c_void_p(parts_dev_ptr), int(n_parts), _ptr(self._u)))
        return self._u.copy()

    # ---- sharded MPPI over peer-to-peer stores (include/ctk_hip.h: ctk_p2p_*) -------------------------
    def p2p_alloc(self, rank: int, world: int) -> bytes:
        buf = C.create_string_buffer(64)
        self._check(self._lib.ctk_p2p_alloc(self._h, int(rank), int(world), buf))
        return bytes(buf.raw)

    def p2p_connect(self, handles) -> None:
        blob = b"".join(bytes(h) for h in handles)
        self._check(self._lib.ctk_p2p_connect(self._h, C.c_char_p(blob)))

    def p2p_step(self, s, samples=None, u_prev=None) -> np.ndarray:
        s = _f32(s).reshape(-1)
        up = None if u_prev is None else _f32(u_prev).reshape(-1)[:self.C].copy()
        if samples is None:
            sp, loc = None, LOC_NONE
        elif isinstance(samples, int):
            sp, loc = C.c_void_p(samples), LOC_DEVICE
        else:
            arr = _f32(samples); sp, loc = _ptr(arr), LOC_HOST
        self._check(self._lib.ctk_p2p_step(self._h, _ptr(s), _ptr(up), sp, loc, _ptr(self._u)))
        return self._u.copy()

    def p2p_close(self) -> None:
        self._check(self._lib.ctk_p2p_close(self._h))

    # ---- sharded CEM / random-action ---------------------------------------------------------------
    def shard_candidates_size(self) -> int:
        return int(self._lib.ctk_shard_candidates_size(self._h))

    def shard_iterations(self) -> int:
        return int(self._lib.ctk_shard_iterations(self._h))

    def shard_iter_begin(self, s, cand_dev_ptr: int, samples=None, u_prev=None):
        self._s[:] = np.asarray(s).reshape(-1)
        up_p = None
        if u_prev is not None:
            self._up[:] = np.asarray(u_prev).reshape(-1)[:self.C]
            up_p = self._up_p
        if samples is None:
            sp, loc = None, LOC_NONE
        elif type(samples) is int:
            sp, loc = samples, LOC_DEVICE
        else:
            arr = _f32(samples)
            if arr.size != self.N * self.H * self.C:
                raise ValueError(f"one iteration consumes {self.N * self.H * self.C} draws, got {arr.size}")
            sp, loc = arr.ctypes.data, LOC_HOST
        self._check(self._lib.ctk_shard_iter_begin(self._h, self._s_p, up_p, sp, loc, cand_dev_ptr))

    def shard_iter_end(self, cands_all_ptr: int, n_ranks: int):
        self._check(self._lib.ctk_shard_iter_end(self._h, cands_all_ptr, int(n_ranks)))

    def shard_finish(self) -> np.ndarray:
        self._check(self._lib.ctk_shard_finish(self._h, self._u_p))
        return self._u.copy()

    # ---- sharded RPGD ---------------------------------------------------------------------------------
    def rpgd_keepers_size(self) -> int:
        return int(self._lib.ctk_rpgd_keepers_size(self._h))

    def rpgd_fresh_rows(self, n_ranks: int) -> int:
        return int(self._lib.ctk_rpgd_fresh_rows(self._h, int(n_ranks)))

    def rpgd_step_begin(self, s, keep_dev_ptr: int, u_prev=None):
        self._s[:] = np.asarray(s).reshape(-1)
        up_p = None
        if u_prev is not None:
            self._up[:] = np.asarray(u_prev).reshape(-1)[:self.C]
            up_p = self._up_p
        self._check(self._lib.ctk_rpgd_step_begin(self._h, self._s_p, up_p, keep_dev_ptr))

    def rpgd_step_end(self, keep_all_ptr: int, n_ranks: int, draws=None) -> np.ndarray:
        if draws is None:
            dp, loc = None, LOC_NONE
        elif type(draws) is int:
            dp, loc = draws, LOC_DEVICE
        else:
            arr = _f32(draws); dp, loc = arr.ctypes.data, LOC_HOST
        self._check(self._lib.ctk_rpgd_step_end(self._h, keep_all_ptr, int(n_ranks), dp, loc, self._u_p))
        return self._u.copy()

    def read(self, name: str) -> np.ndarray:
        N, H, S, Cn = self.N, self.H, self.S, self.C
        cap = max(N * (H + 1) * S, N * H * Cn, 1)
        buf = np.empty(cap, np.float32)
        n = C.c_size_t()
        self._check(self._lib.ctk_read(self._h, BUFFERS[name], _ptr(buf), cap, C.byref(n)))
        out = buf[: n.value].copy()
        shapes = {"Q": (N, H, Cn), "J": (N,), "TRAJ": (N, H + 1, S), "U_NOM": (1, H, Cn), "STD": (1, H, Cn),
                  "ADAM_M": (N, H, Cn), "ADAM_V": (N, H, Cn), "AGES": (N,), "PLAN": (N, H, Cn), "AGES_LOGGED": (N,)}
        if name == "BEST_IDX":
            return out.astype(np.int64)
        return out.reshape(shapes[name])

    # resident MPPI step (include/ctk_hip.h: ctk_resident_*): the first step launches a kernel that stays on the device and serves the
    # following steps from a pinned mailbox; it leaves by itself after idle_us without a request, and at once on resident_stop() or any
    # other call that touches device state
    def resident_enable(self, on: bool = True, idle_us: float = 200.0, read_ahead: bool = False):
        """read_ahead: the caller's promise that the device sample buffers it hands to step() keep their contents while the resident
        form is enabled (a static pool): they are then read between steps.  Without it a buffer is read when its request arrives, so
        refilling one buffer in place between steps is safe."""
        self._check(self._lib.ctk_resident_enable(self._h, (2 if read_ahead else 1) if on else 0, float(idle_us)))

    def resident_stop(self):
        self._check(self._lib.ctk_resident_stop(self._h))

    def resident_stats(self) -> dict:
        a, b, r, m = C.c_uint64(0), C.c_uint64(0), C.c_int(0), C.c_int(0)
        self._check(self._lib.ctk_resident_stats(self._h, C.byref(a), C.byref(b), C.byref(r), C.byref(m)))
        return {"launches": int(a.value), "steps": int(b.value), "running": bool(r.value), "mailbox": "device memory" if m.value else "pinned host memory"}

    # device-resident step log (include/ctk_hip.h: ctk_log_*)
    def log_enable(self, capacity_steps: int):
        self._check(self._lib.ctk_log_enable(self._h, int(capacity_steps)))
        self.log_capacity = int(capacity_steps)

    def log_count(self) -> int:
        return int(self._lib.ctk_log_count(self._h))

    def log_read(self, name: str, first_step: int, n_steps: int) -> np.ndarray:
        N, H = self.N, self.H
        shape = {"Q": (N, H, self.C), "J": (N,), "TRAJ": (N, H + 1, self.S), "AGES": (N,)}[name]
        out = np.empty((int(n_steps),) + shape, np.float32)
        n = C.c_size_t()
        self._check(self._lib.ctk_log_read(self._h, BUFFERS[name], int(first_step), int(n_steps), _ptr(out), out.size, C.byref(n)))
        assert n.value == out.size
        return out

    def get_state(self) -> np.ndarray:
        n = int(self._lib.ctk_state_size(self._h))
        buf = np.empty(n, np.float32)
        self._check(self._lib.ctk_get_state(self._h, _ptr(buf), n))
        return buf

    def set_state(self, state):
        st = _f32(state).ravel()
        self._check(self._lib.ctk_set_state(self._h, _ptr(st), st.size))

    def profile_enable(self, on=True, every: int = 1):
        """time every `every`-th launch of the dominant kernel (dispatch timestamps); on=False disables"""
        self._check(self._lib.ctk_profile_enable(self._h, int(every) if on else 0))

    def profile_read(self) -> np.ndarray:
        buf = np.empty(4096, np.float32)
        n = C.c_size_t()
        self._check(self._lib.ctk_profile_read(self._h, _ptr(buf), buf.size, C.byref(n)))
        return buf[: n.value].copy()

    def dominant_kernel(self) -> str:
        return self._lib.ctk_dominant_kernel(self._h).decode()
