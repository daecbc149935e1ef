"""TEST / MEASUREMENT INFRASTRUCTURE ONLY — ctypes binding of oracle/ctk_cpu.c, the native multi-core (C + OpenMP) restatement of the MPPI
and random-action steps on the CartPole analytic predictor (see the C file's header for the reference lines it follows).  Users: tests/
(held against the reference-recorded goldens and the NumPy oracle) and bench.py's `cpu_baseline` leg.  Never imported by the product.

    build()                      gcc -O3 -ffp-contract=off -fopenmp -shared -fPIC -> oracle/_build/libctk_cpu.so  (no -march=native: the file
                                 is built in the build container and travels to the GPU box, whose host CPU may differ)
    MppiCpu(env, dt, ...)        .step(s, noise[N,P]) -> u        (state: u_nom[H], u)
    random_action_step(...)      -> (u, J[N], best)
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import ctk_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "ctk_cpu.c")
LIB = os.path.join(HERE, "_build", "libctk_cpu.so")
_lib = None


class _Env(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("dt", "u_max", "g", "M_fric", "inv_mt", "k_ml", "k_jf", "k43l", "k_mpl_mt", "inv_xs", "ep_c", "ccR",
                                         "target_position", "dd_weight", "ekp_weight", "ccrc_weight", "terminal_weight")] + [("intermediate_steps", C.c_int)]


def build(force: bool = False) -> str:
    """compile the C restatement (no -ffast-math, no FMA contraction: the NumPy oracle it follows does not fuse either)"""
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["gcc", "-O3", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"], check=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        l = C.CDLL(LIB)
        l.ctkc_abi.restype = C.c_int
        l.ctkc_env_size.restype = C.c_int
        l.ctkc_max_threads.restype = C.c_int
        if l.ctkc_abi() != 1 or l.ctkc_env_size() != C.sizeof(_Env):
            raise RuntimeError("oracle/_build/libctk_cpu.so does not match oracle/ctk_cpu.py (rebuild: oracle.ctk_cpu.build(force=True))")
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        l.ctkc_mppi_step.argtypes = [C.POINTER(_Env), C.c_int, C.c_int, C.c_int, ip, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_float, fp, C.c_float, fp, fp, fp, fp, C.c_int]
        l.ctkc_mppi_step.restype = C.c_int
        l.ctkc_random_action_step.argtypes = [C.POINTER(_Env), C.c_int, C.c_int, C.c_float, C.c_float, fp, C.c_float, fp, fp, ip, fp, C.c_int]
        l.ctkc_random_action_step.restype = C.c_int
        _lib = l
    return _lib


def max_threads() -> int:
    return int(lib().ctkc_max_threads())


def _env_struct(env: "O.EnvParams", dt: float, intermediate_steps: int = 1) -> _Env:
    k = O.derived_constants(env, dt, intermediate_steps)
    e = _Env()
    for n in ("dt", "u_max", "g", "M_fric", "inv_mt", "k_ml", "k_jf", "k43l", "k_mpl_mt", "inv_xs", "ep_c", "ccR"):
        setattr(e, n, float(k["k_mpl_mt" if n == "k_mpl_mt" else n]))
    for n in ("target_position", "dd_weight", "ekp_weight", "ccrc_weight", "terminal_weight"):
        setattr(e, n, float(np.float32(getattr(env, n))))
    e.intermediate_steps = int(intermediate_steps)
    return e


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class MppiCpu:
    """the C MPPI step with the oracle MPPI's constructor arguments (CartPole, analytic predictor, one control input)"""

    def __init__(self, env=None, dt=0.02, low=-1.0, high=1.0, *, num_rollouts, mpc_horizon, cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0,
                 SQRTRHOINV=0.03, period_interpolation_inducing_points=10, intermediate_steps=1, threads=None):
        self.env = env if env is not None else O.EnvParams()
        self.N, self.H = int(num_rollouts), int(mpc_horizon)
        self.P = O.num_inducing_points(self.H, period_interpolation_inducing_points)
        i0, w0, w1 = O.interpolation_table(self.H, period_interpolation_inducing_points)
        self.i0 = np.ascontiguousarray(i0, np.int32); self.w0 = np.ascontiguousarray(w0, np.float32); self.w1 = np.ascontiguousarray(w1, np.float32)
        self.k = _env_struct(self.env, dt, intermediate_steps)
        self.lo, self.hi = float(np.float32(low)), float(np.float32(high))
        self.cc_weight, self.R, self.NU, self.LBD = float(np.float32(cc_weight)), float(np.float32(R)), float(np.float32(NU)), float(LBD)
        self.stdev = float(O.f32(np.array(SQRTRHOINV) * (1 / np.sqrt(dt))))                 # optimizer_mppi.py:130
        self.threads = int(threads) if threads else max_threads()
        self.u_nom = np.full(self.H, np.float32(0.5) * (np.float32(low) + np.float32(high)), np.float32)   # :227-231
        self.u = np.float32(0.0)
        self.J = np.empty(self.N, np.float32)
        self._u_out = np.zeros(1, np.float32)

    def step(self, s, noise):
        s = np.ascontiguousarray(s, np.float32).reshape(4)
        noise = np.ascontiguousarray(noise, np.float32).reshape(self.N, self.P)
        rc = lib().ctkc_mppi_step(C.byref(self.k), self.N, self.H, self.P, self.i0.ctypes.data_as(C.POINTER(C.c_int)), _fp(self.w0), _fp(self.w1),
                                  self.stdev, self.lo, self.hi, self.cc_weight, self.R, self.NU, self.LBD, _fp(s), float(self.u), _fp(noise),
                                  _fp(self.u_nom), _fp(self.J), _fp(self._u_out), self.threads)
        if rc != 0:
            raise MemoryError("ctkc_mppi_step")
        self.u = np.float32(self._u_out[0])
        return np.array([self.u], np.float32)


def random_action_step(env, dt, low, high, s, u_prev, u01, threads=None):
    """-> (u, J[N], best index) for U[0,1) draws u01 [N, H]"""
    u01 = np.ascontiguousarray(u01, np.float32)
    N, H = u01.shape[0], u01.shape[1]
    k = _env_struct(env, dt, 1)
    J = np.empty(N, np.float32)
    best = C.c_int(0)
    u = np.zeros(1, np.float32)
    s = np.ascontiguousarray(s, np.float32).reshape(4)
    lib().ctkc_random_action_step(C.byref(k), N, H, float(np.float32(low)), float(np.float32(high)), _fp(s), float(np.float32(u_prev)),
                                  _fp(u01.reshape(N, H)), _fp(J), C.byref(best), _fp(u), int(threads) if threads else max_threads())
    return np.float32(u[0]), J, int(best.value)
