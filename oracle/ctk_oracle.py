"""CPU oracle for the sampling-based MPC hot path (MPPI / CEM / RPGD / random-action).

TEST INFRASTRUCTURE ONLY.  Nothing in the product (`control_toolkit_amd/`) may import this
module; only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do,
and there only as the checker / the timed CPU baseline.

What it is: a plain NumPy fp32 restatement of the optimizer logic of SensorsINI/Control_Toolkit
(reference @ 2025-11-28).  Every function cites the reference file:line it follows.

Pinning status (see DESIGN.md "Oracle"):
  * Interpolator, cost aggregation (mean over H+1), MPPI step, the torch-branch ADAM and the
    RPGD step are PINNED against the unmodified reference modules executed in the build
    container (tests/golden/make_golden.py -> tests/golden/*.npz).
  * CEM and random-action cannot be imported (module-level `import tensorflow`); they follow
    the source text of optimizer_cem_tf.py / optimizer_random_action_tf.py: parity unpinned by
    any reference execution.
  * The predictor (cart-pole ODE step, 5-32-32-4 tanh MLP) and the concrete stage/terminal cost
    live in repositories that are not vendored with the reference (SI_Toolkit,
    Control_Toolkit_ASF): they are defined by this build, parity unpinned.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, asdict
from typing import Optional, Tuple

import numpy as np

f32 = np.float32

# ----------------------------------------------------------------------------------------------
# Parameters (environment + cost).  Ids mirror include/ctk_hip.h : enum ctk_param.
# ----------------------------------------------------------------------------------------------
PARAM_NAMES = (
    "g", "m_cart", "m_pole", "L", "u_max", "M_fric", "J_fric",           # dynamics
    "target_position", "target_equilibrium",                             # per-step attributes
    "dd_weight", "ep_weight", "ekp_weight", "cc_weight", "ccrc_weight",  # cost weights
    "R", "x_scale", "terminal_weight",
)


@dataclass
class EnvParams:
    NAME = "CartPole"
    S, C = 4, 1
    """Physical + cost parameters of the build-defined cart-pole (state order position,
    positionD, angle, angleD as hinted by reference Controllers/controller_C.py:14-19; weight
    names as in Control_Toolkit_ASF_Template/config_cost_function.yml:11-18; per-step
    attributes as seeded by controller_server/controller_server.py:21-28)."""
    g: float = 9.81
    m_cart: float = 0.230
    m_pole: float = 0.087
    L: float = 0.1975            # half length of the pole [m]
    u_max: float = 2.62          # force at Q = 1 [N]
    M_fric: float = 4.77         # cart friction [N s / m]
    J_fric: float = 2.5e-4       # joint friction [N m s]
    target_position: float = 0.0
    target_equilibrium: float = 1.0
    dd_weight: float = 600.0
    ep_weight: float = 20000.0
    ekp_weight: float = 80.0
    cc_weight: float = 1.0
    ccrc_weight: float = 1.0
    R: float = 1.0
    x_scale: float = 0.198       # track half length used to normalise the distance term
    terminal_weight: float = 0.0

    def as_array(self) -> np.ndarray:
        return np.array([getattr(self, k) for k in PARAM_NAMES], dtype=np.float32)

    def param_names(self):
        return PARAM_NAMES


# Second build-defined environment ("a 2-input, 6-state system of your choosing"): a planar
# quadrotor.  State (x, vx, z, vz, theta, omega), inputs (u1, u2) in [-1, 1] = normalised rotor commands, rotor
# thrust T_i = (m g / 2) (1 + thrust_gain u_i).  Selected by environment_name / predictor_specification /
# cost_function_specification exactly like the reference selects its plants (controller_mpc.py:67-82,
# cost_function_wrapper.py:59-66).  Parity unpinned by nature (the reference's environments are not vendored).
QUAD2D_PARAM_NAMES = (
    "g", "mass", "inertia", "arm", "thrust_gain", "drag_lin", "drag_ang",     # dynamics
    "target_x", "target_z",                                                    # per-step attributes
    "pos_weight", "ang_weight", "vel_weight", "angvel_weight", "cc_weight", "ccrc_weight",
    "R", "pos_scale", "terminal_weight",
)


@dataclass
class Quad2DParams:
    NAME = "Quad2D"
    S, C = 6, 2
    g: float = 9.81
    mass: float = 0.5
    inertia: float = 0.004
    arm: float = 0.12
    thrust_gain: float = 0.6
    drag_lin: float = 0.25
    drag_ang: float = 0.4
    target_x: float = 0.0
    target_z: float = 1.0
    pos_weight: float = 400.0
    ang_weight: float = 150.0
    vel_weight: float = 8.0
    angvel_weight: float = 1.5
    cc_weight: float = 1.0
    ccrc_weight: float = 2.0
    R: float = 1.0
    pos_scale: float = 0.5
    terminal_weight: float = 0.0

    def as_array(self) -> np.ndarray:
        return np.array([getattr(self, k) for k in QUAD2D_PARAM_NAMES], dtype=np.float32)

    def param_names(self):
        return QUAD2D_PARAM_NAMES


# Third build-defined environment (more than 8 network inputs, a third layer-1 k-step): a planar hovercraft
# with a reaction wheel.  State (x, vx, y, vy, theta, omega, wheel speed), inputs (u1, u2, u3) in [-1, 1] = main thruster
# (body axis), lateral thruster, wheel torque; the wheel's torque reacts on the body.  7 states + 3 inputs = 10 network inputs.
HOVER_PARAM_NAMES = (
    "mass", "inertia", "wheel_inertia", "thrust_max", "lateral_max", "torque_max", "drag_lin", "drag_ang", "wheel_friction",   # dynamics
    "target_x", "target_y",                                                                                                  # per-step attributes
    "pos_weight", "ang_weight", "vel_weight", "angvel_weight", "wheel_weight", "cc_weight", "ccrc_weight",
    "R", "pos_scale", "terminal_weight",
)


@dataclass
class HoverParams:
    NAME = "Hover"
    S, C = 7, 3
    mass: float = 1.2
    inertia: float = 0.05
    wheel_inertia: float = 0.01
    thrust_max: float = 4.0
    lateral_max: float = 1.5
    torque_max: float = 0.2
    drag_lin: float = 0.3
    drag_ang: float = 0.2
    wheel_friction: float = 0.05
    target_x: float = 0.0
    target_y: float = 0.0
    pos_weight: float = 300.0
    ang_weight: float = 80.0
    vel_weight: float = 6.0
    angvel_weight: float = 1.0
    wheel_weight: float = 0.02
    cc_weight: float = 1.0
    ccrc_weight: float = 1.5
    R: float = 1.0
    pos_scale: float = 0.5
    terminal_weight: float = 0.0

    def as_array(self) -> np.ndarray:
        return np.array([getattr(self, k) for k in HOVER_PARAM_NAMES], dtype=np.float32)

    def param_names(self):
        return HOVER_PARAM_NAMES


ENVIRONMENTS = {"CartPole": EnvParams, "Quad2D": Quad2DParams, "Hover": HoverParams}


def hover_constants(p: HoverParams, dt: float, intermediate_steps: int = 1) -> dict:
    """Derived fp32 constants of the hovercraft (double, rounded once) — the same expressions as csrc/ctk_env.h:
    Env<CTK_ENV_HOVER>::derive."""
    q = {k: float(np.float32(getattr(p, k))) for k in HOVER_PARAM_NAMES}
    d = dict(
        dt=dt / intermediate_steps,
        aF=q["thrust_max"] / q["mass"], aL=q["lateral_max"] / q["mass"],
        kT=q["torque_max"] / q["inertia"], kW=q["torque_max"] / q["wheel_inertia"],
        c_v=q["drag_lin"], c_w=q["drag_ang"], c_ww=q["wheel_friction"],
        pos_c=q["pos_weight"] / (q["pos_scale"] * q["pos_scale"]),
        ccR=q["cc_weight"] * q["R"],
    )
    return {k: np.float32(v) for k, v in d.items()}


def quad2d_constants(p: Quad2DParams, dt: float, intermediate_steps: int = 1) -> dict:
    """Derived fp32 constants of the planar quadrotor (double, rounded once) — the same expressions as
    csrc/ctk_env.h: Env<CTK_ENV_QUAD2D>::derive."""
    q = {k: float(np.float32(getattr(p, k))) for k in QUAD2D_PARAM_NAMES}
    d = dict(
        dt=dt / intermediate_steps,
        g=q["g"],
        kF=0.5 * q["g"] * q["thrust_gain"],                                        # (T1+T2)/m = g + kF (u1+u2)
        kM=q["arm"] * 0.5 * q["mass"] * q["g"] * q["thrust_gain"] / q["inertia"],  # arm (T1-T2)/I = kM (u1-u2)
        c_v=q["drag_lin"], c_w=q["drag_ang"],
        pos_c=q["pos_weight"] / (q["pos_scale"] * q["pos_scale"]),
        ccR=q["cc_weight"] * q["R"],
    )
    return {k: np.float32(v) for k, v in d.items()}


def derived_constants(p, dt: float, intermediate_steps: int = 1) -> dict:
    """Constants the step function uses, computed in double then rounded once to fp32.
    The C library computes exactly the same expressions (csrc/ctk_common.h: derive_constants)."""
    if isinstance(p, Quad2DParams):
        return quad2d_constants(p, dt, intermediate_steps)
    if isinstance(p, HoverParams):
        return hover_constants(p, dt, intermediate_steps)
    q = {k: float(np.float32(getattr(p, k))) for k in PARAM_NAMES}  # primary params are fp32
    inv_mt = 1.0 / (q["m_cart"] + q["m_pole"])
    ml = q["m_pole"] * q["L"]
    d = dict(
        dt=dt / intermediate_steps,
        u_max=q["u_max"],
        g=q["g"],
        M_fric=q["M_fric"],
        inv_mt=inv_mt,
        k_ml=ml,
        k_jf=q["J_fric"] / ml,
        k43l=q["L"] * (4.0 / 3.0),
        k_mpl_mt=ml * inv_mt,
        inv_xs=1.0 / q["x_scale"],
        ep_c=q["ep_weight"] * q["target_equilibrium"] * 0.25,
        ccR=q["cc_weight"] * q["R"],
    )
    return {k: np.float32(v) for k, v in d.items()}


# ----------------------------------------------------------------------------------------------
# Predictors (build-defined; reference call sites: optimizer_mppi.py:188, optimizer_cem_tf.py:57,
# optimizer_rpgd.py:300, optimizer_random_action_tf.py:42 -> predictor.predict_core(s, Q))
# ----------------------------------------------------------------------------------------------
MLP_IN, MLP_H, MLP_OUT = 5, 32, 4
MLP_NUM_WEIGHTS = MLP_IN * MLP_H + MLP_H + MLP_H * MLP_H + MLP_H + MLP_H * MLP_OUT + MLP_OUT  # 1380


def mlp_num_weights(n_in: int = MLP_IN, n_out: int = MLP_OUT, hidden=(MLP_H, MLP_H)) -> int:
    h1, h2 = hidden
    return n_in * h1 + h1 + h1 * h2 + h2 + h2 * n_out + n_out


def mlp_default_weights(seed: int = 0, n_in: int = MLP_IN, n_out: int = MLP_OUT, hidden=(MLP_H, MLP_H)) -> np.ndarray:
    """(S+C)->h1->h2->S tanh MLP (cart-pole default: 5->32->32->4; the reference names a network by its sizes,
    `Dense-<I>IN-<h1>H1-<h2>H2-<O>OUT-<n>`, Control_Toolkit_ASF_Template/config_controllers.yml:8), weights N(0, 1/fan_in), biases
    N(0, 0.01) from default_rng(seed).  Flat layout: W1[h1,n_in] b1[h1] W2[h2,h1] b2[h2] W3[n_out,h2] b3[n_out]."""
    h1, h2 = hidden
    rng = np.random.default_rng(seed)
    W1 = rng.normal(0, 1 / math.sqrt(n_in), (h1, n_in))
    b1 = rng.normal(0, 0.1, (h1,))
    W2 = rng.normal(0, 1 / math.sqrt(h1), (h2, h1))
    b2 = rng.normal(0, 0.1, (h2,))
    W3 = rng.normal(0, 1 / math.sqrt(h2), (n_out, h2))
    b3 = rng.normal(0, 0.1, (n_out,))
    return np.concatenate([a.ravel() for a in (W1, b1, W2, b2, W3, b3)]).astype(np.float32)


def mlp_unpack(w: np.ndarray, n_in: int = MLP_IN, n_out: int = MLP_OUT, hidden=(MLP_H, MLP_H)):
    w = np.asarray(w, dtype=np.float32)
    h1, h2 = hidden
    assert w.size == mlp_num_weights(n_in, n_out, hidden)
    o = 0
    def take(n, shape):
        nonlocal o
        a = w[o:o + n].reshape(shape)
        o += n
        return a
    W1 = take(h1 * n_in, (h1, n_in)); b1 = take(h1, (h1,))
    W2 = take(h2 * h1, (h2, h1)); b2 = take(h2, (h2,))
    W3 = take(n_out * h2, (n_out, h2)); b3 = take(n_out, (n_out,))
    return W1, b1, W2, b2, W3, b3


# GRU predictor (build-defined; the reference only hints at it through the network-name convention
# 'GRU-6IN-32H1-32H2-5OUT-0', Control_Toolkit_ASF_Template/config_controllers.yml:8, and through
# predictor.update(s, Q0) — the hidden-state advance at optimizer_mppi.py:195-197).  Two GRU layers of
# 32 units (PyTorch gate convention: r, z, n; reset applied after the recurrent matmul) + dense 32->4:
#   r = sig(W_ir x + b_ir + W_hr h + b_hr);  z = sig(W_iz x + b_iz + W_hz h + b_hz)
#   n = tanh(W_in x + b_in + r * (W_hn h + b_hn));  h' = (1 - z) * n + z * h
# Flat layout: per layer W_i[96,I] W_h[96,32] b_i[96] b_h[96] (rows r|z|n), then W_o[4,32] b_o[4].
GRU_H = 32
GRU_NUM_WEIGHTS = (96 * 5 + 96 * 32 + 192) + (96 * 32 + 96 * 32 + 192) + (4 * 32 + 4)   # 10212


def gru_num_weights(n_in: int = MLP_IN, n_out: int = MLP_OUT) -> int:
    return (96 * n_in + 96 * 32 + 192) + (96 * 32 + 96 * 32 + 192) + (n_out * 32 + n_out)


def gru_default_weights(seed: int = 0, n_in: int = MLP_IN, n_out: int = MLP_OUT) -> np.ndarray:
    rng = np.random.default_rng(seed)
    parts = []
    for I in (n_in, GRU_H):
        parts += [rng.normal(0, 1 / math.sqrt(I), (96, I)), rng.normal(0, 1 / math.sqrt(GRU_H), (96, GRU_H)),
                  rng.normal(0, 0.1, (96,)), rng.normal(0, 0.1, (96,))]
    parts += [rng.normal(0, 1 / math.sqrt(GRU_H), (n_out, GRU_H)), rng.normal(0, 0.1, (n_out,))]
    return np.concatenate([a.ravel() for a in parts]).astype(np.float32)


def gru_unpack(w: np.ndarray, n_in: int = MLP_IN, n_out: int = MLP_OUT):
    w = np.asarray(w, dtype=np.float32)
    assert w.size == gru_num_weights(n_in, n_out)
    o = 0
    layers = []
    for I in (n_in, GRU_H):
        Wi = w[o:o + 96 * I].reshape(96, I); o += 96 * I
        Wh = w[o:o + 96 * GRU_H].reshape(96, GRU_H); o += 96 * GRU_H
        bi = w[o:o + 96]; o += 96
        bh = w[o:o + 96]; o += 96
        layers.append((Wi, Wh, bi, bh))
    Wo = w[o:o + n_out * GRU_H].reshape(n_out, GRU_H); o += n_out * GRU_H
    bo = w[o:o + n_out]
    return layers, Wo, bo


def _sigmoid(x):
    return (f32(1.0) / (f32(1.0) + np.exp(-x))).astype(np.float32)


def gru_cell(x, h, Wi, Wh, bi, bh):
    gi = (x @ Wi.T + bi).astype(np.float32)
    gh = (h @ Wh.T + bh).astype(np.float32)
    r = _sigmoid(gi[:, :32] + gh[:, :32])
    z = _sigmoid(gi[:, 32:64] + gh[:, 32:64])
    n = np.tanh(gi[:, 64:] + r * gh[:, 64:]).astype(np.float32)
    return ((f32(1.0) - z) * n + z * h).astype(np.float32)


def gru_cell_fwd(x, h, Wi, Wh, bi, bh):
    """gru_cell + what its adjoint needs"""
    gi = (x @ Wi.T + bi).astype(np.float32)
    gh = (h @ Wh.T + bh).astype(np.float32)
    r = _sigmoid(gi[:, :32] + gh[:, :32])
    z = _sigmoid(gi[:, 32:64] + gh[:, 32:64])
    ghn = gh[:, 64:]
    n = np.tanh(gi[:, 64:] + r * ghn).astype(np.float32)
    return ((f32(1.0) - z) * n + z * h).astype(np.float32), (x, h, r, z, n, ghn)


def gru_cell_bwd(dh_new, cache, Wi, Wh):
    """vector-Jacobian product of one GRU cell: adjoint of h' -> (adjoint of x, adjoint of h)"""
    x, h, r, z, n, ghn = cache
    dn = dh_new * (f32(1.0) - z)
    dz = dh_new * (h - n)
    dh = dh_new * z
    dan = dn * (f32(1.0) - n * n)            # pre-activation of n: gi_n + r * gh_n
    dr = dan * ghn
    dghn = dan * r
    daz = dz * z * (f32(1.0) - z)
    dar = dr * r * (f32(1.0) - r)
    dgi = np.concatenate([dar, daz, dan], 1).astype(np.float32)
    dgh = np.concatenate([dar, daz, dghn], 1).astype(np.float32)
    return (dgi @ Wi).astype(np.float32), (dh + dgh @ Wh).astype(np.float32)


@dataclass
class Predictor:
    """kind = "ODE" (analytic cart-pole, explicit Euler), "MLP" (direct next-state net) or "GRU"
    (recurrent net; `hidden` [2,32] is the state every rollout starts from, advanced by update())."""
    kind: str = "ODE"
    dt: float = 0.02
    intermediate_steps: int = 1
    env: EnvParams = field(default_factory=EnvParams)
    weights: Optional[np.ndarray] = None
    hidden_sizes: tuple = (MLP_H, MLP_H)     # MLP: the two hidden widths (the <h1>H1-<h2>H2 of the network name)

    def __post_init__(self):
        self.S, self.C = self.env.S, self.env.C
        if self.kind == "MLP" and self.weights is None:
            self.weights = mlp_default_weights(0, self.S + self.C, self.S, self.hidden_sizes)
        if self.kind == "GRU":
            if self.weights is None:
                self.weights = gru_default_weights(0, self.S + self.C, self.S)
            self.hidden = np.zeros((2, GRU_H), np.float32)

    def _q2(self, q):
        """inputs as [N,C] (callers of the C = 1 environment may pass [N])"""
        q = np.asarray(q, np.float32)
        return q.reshape(-1, self.C) if q.ndim != 2 else q

    # GRU -------------------------------------------------------------------------------------
    def _gru_step(self, s, q, h1, h2):
        layers, Wo, bo = gru_unpack(self.weights, self.S + self.C, self.S)
        xin = np.concatenate([s, self._q2(q)], axis=1).astype(np.float32)
        h1n = gru_cell(xin, h1, *layers[0])
        h2n = gru_cell(h1n, h2, *layers[1])
        return (h2n @ Wo.T + bo).astype(np.float32), h1n, h2n

    def update(self, s, q0):
        """predictor.update(s, Q0) (optimizer_mppi.py:195-197): advance the carried hidden state by the
        real state and the applied input."""
        if self.kind != "GRU":
            return
        s = np.asarray(s, np.float32).reshape(1, self.S)
        _, h1, h2 = self._gru_step(s, np.asarray(q0, np.float32).reshape(1, self.C), self.hidden[0:1], self.hidden[1:2])
        self.hidden = np.concatenate([h1, h2], 0)

    # one predictor step ---------------------------------------------------------------------
    def step(self, s: np.ndarray, q: np.ndarray) -> np.ndarray:
        """s [N,S] fp32, q [N,C] fp32 ([N] accepted when C = 1) -> next state [N,S] fp32."""
        if self.kind == "ODE":
            return self._ode_step(s, q)
        return self._mlp_step(s, q)[0]

    def _quad_step(self, s, q):
        k = derived_constants(self.env, self.dt, self.intermediate_steps)
        x, vx, z, vz, th, om = (s[:, i].copy() for i in range(6))
        q = self._q2(q)
        aF = k["g"] + k["kF"] * (q[:, 0] + q[:, 1])        # total thrust / mass
        aM = k["kM"] * (q[:, 0] - q[:, 1])                 # rotor torque / inertia
        dt = k["dt"]
        for _ in range(self.intermediate_steps):
            sn, cs = np.sin(th), np.cos(th)
            ax = -aF * sn - k["c_v"] * vx
            az = aF * cs - k["g"] - k["c_v"] * vz
            al = aM - k["c_w"] * om
            x, vx, z, vz, th, om = (x + dt * vx, vx + dt * ax, z + dt * vz, vz + dt * az, th + dt * om, om + dt * al)
        return np.stack([x, vx, z, vz, th, om], axis=1).astype(np.float32)

    def _hover_step(self, s, q):
        k = derived_constants(self.env, self.dt, self.intermediate_steps)
        x, vx, y, vy, th, om, w = (s[:, i].copy() for i in range(7))
        q = self._q2(q)
        fb, fl = k["aF"] * q[:, 0], k["aL"] * q[:, 1]     # body-frame accelerations of the two thrusters
        dt = k["dt"]
        for _ in range(self.intermediate_steps):
            sn, cs = np.sin(th), np.cos(th)
            ax = fb * cs - fl * sn - k["c_v"] * vx
            ay = fb * sn + fl * cs - k["c_v"] * vy
            al = -k["kT"] * q[:, 2] - k["c_w"] * om        # the wheel's torque reacts on the body
            aw = k["kW"] * q[:, 2] - k["c_ww"] * w
            x, vx, y, vy, th, om, w = (x + dt * vx, vx + dt * ax, y + dt * vy, vy + dt * ay, th + dt * om, om + dt * al, w + dt * aw)
        return np.stack([x, vx, y, vy, th, om, w], axis=1).astype(np.float32)

    def _hover_vjp(self, s, q, lam):
        assert self.intermediate_steps == 1, "adjoint restated for intermediate_steps == 1"
        k = derived_constants(self.env, self.dt, 1)
        q = self._q2(q)
        th = s[:, 4]
        lx, lvx, ly, lvy, lth, lom, lw = (lam[:, i] for i in range(7))
        dt = k["dt"]
        sn, cs = np.sin(th), np.cos(th)
        fb, fl = k["aF"] * q[:, 0], k["aL"] * q[:, 1]
        a_ax, a_ay, a_al, a_aw = dt * lvx, dt * lvy, dt * lom, dt * lw
        o_x = lx
        o_vx = lvx + dt * lx - k["c_v"] * a_ax
        o_y = ly
        o_vy = lvy + dt * ly - k["c_v"] * a_ay
        o_th = lth + (-fb * sn - fl * cs) * a_ax + (fb * cs - fl * sn) * a_ay
        o_om = lom + dt * lth - k["c_w"] * a_al
        o_w = lw - k["c_ww"] * a_aw
        g0 = k["aF"] * (cs * a_ax + sn * a_ay)
        g1 = k["aL"] * (-sn * a_ax + cs * a_ay)
        g2 = -k["kT"] * a_al + k["kW"] * a_aw
        return (np.stack([o_x, o_vx, o_y, o_vy, o_th, o_om, o_w], 1).astype(np.float32),
                np.stack([g0, g1, g2], 1).astype(np.float32))

    def _quad_vjp(self, s, q, lam):
        assert self.intermediate_steps == 1, "adjoint restated for intermediate_steps == 1"
        k = derived_constants(self.env, self.dt, 1)
        q = self._q2(q)
        vx, vz, th, om = s[:, 1], s[:, 3], s[:, 4], s[:, 5]
        lx, lvx, lz, lvz, lth, lom = (lam[:, i] for i in range(6))
        dt = k["dt"]
        sn, cs = np.sin(th), np.cos(th)
        aF = k["g"] + k["kF"] * (q[:, 0] + q[:, 1])
        a_ax, a_az, a_al = dt * lvx, dt * lvz, dt * lom
        a_aF = -sn * a_ax + cs * a_az
        o_x = lx
        o_vx = lvx + dt * lx - k["c_v"] * a_ax
        o_z = lz
        o_vz = lvz + dt * lz - k["c_v"] * a_az
        o_th = lth - aF * (cs * a_ax + sn * a_az)
        o_om = lom + dt * lth - k["c_w"] * a_al
        g0 = k["kF"] * a_aF + k["kM"] * a_al
        g1 = k["kF"] * a_aF - k["kM"] * a_al
        return (np.stack([o_x, o_vx, o_z, o_vz, o_th, o_om], 1).astype(np.float32),
                np.stack([g0, g1], 1).astype(np.float32))

    def _ode_step(self, s, q):
        if isinstance(self.env, Quad2DParams):
            return self._quad_step(s, q)
        if isinstance(self.env, HoverParams):
            return self._hover_step(s, q)
        q = self._q2(q)[:, 0]
        k = derived_constants(self.env, self.dt, self.intermediate_steps)
        x, v, th, om = (s[:, i].copy() for i in range(4))
        F = k["u_max"] * q
        for _ in range(self.intermediate_steps):
            sn, cs = np.sin(th), np.cos(th)
            A = F + k["k_ml"] * om * om * sn - k["M_fric"] * v
            tmp = A * k["inv_mt"]
            D = k["k43l"] - k["k_mpl_mt"] * cs * cs
            Nn = k["g"] * sn - cs * tmp - k["k_jf"] * om
            thdd = Nn / D
            xdd = tmp - k["k_mpl_mt"] * thdd * cs
            x, v, th, om = (x + k["dt"] * v, v + k["dt"] * xdd,
                            th + k["dt"] * om, om + k["dt"] * thdd)
        return np.stack([x, v, th, om], axis=1).astype(np.float32)

    def _mlp_step(self, s, q):
        W1, b1, W2, b2, W3, b3 = mlp_unpack(self.weights, self.S + self.C, self.S, self.hidden_sizes)
        xin = np.concatenate([s, self._q2(q)], axis=1).astype(np.float32)
        h1 = np.tanh(xin @ W1.T + b1).astype(np.float32)
        h2 = np.tanh(h1 @ W2.T + b2).astype(np.float32)
        out = (h2 @ W3.T + b3).astype(np.float32)
        return out, (xin, h1, h2)

    # vector-Jacobian product of one step (used by the RPGD adjoint) ---------------------------
    def step_vjp(self, s, q, lam):
        """Given lam = dL/ds' [N,S], return (dL/ds [N,S], dL/dq [N,C])."""
        if self.kind == "ODE":
            return self._ode_vjp(s, q, lam)
        return self._mlp_vjp(s, q, lam)

    def _ode_vjp(self, s, q, lam):
        if isinstance(self.env, Quad2DParams):
            return self._quad_vjp(s, q, lam)
        if isinstance(self.env, HoverParams):
            return self._hover_vjp(s, q, lam)
        q = self._q2(q)[:, 0]
        assert self.intermediate_steps == 1, "adjoint restated for intermediate_steps == 1"
        k = derived_constants(self.env, self.dt, 1)
        x, v, th, om = (s[:, i] for i in range(4))
        lx, lv, lth, lom = (lam[:, i] for i in range(4))
        dt = k["dt"]
        sn, cs = np.sin(th), np.cos(th)
        F = k["u_max"] * q
        A = F + k["k_ml"] * om * om * sn - k["M_fric"] * v
        tmp = A * k["inv_mt"]
        D = k["k43l"] - k["k_mpl_mt"] * cs * cs
        Nn = k["g"] * sn - cs * tmp - k["k_jf"] * om
        thdd = Nn / D
        a_xdd = dt * lv
        a_thdd = dt * lom - k["k_mpl_mt"] * cs * a_xdd
        a_tmp = a_xdd
        a_cs = -k["k_mpl_mt"] * thdd * a_xdd
        a_Nn = a_thdd / D
        a_D = -a_Nn * thdd
        a_sn = k["g"] * a_Nn
        a_cs = a_cs - tmp * a_Nn - f32(2.0) * k["k_mpl_mt"] * cs * a_D
        a_tmp = a_tmp - cs * a_Nn
        a_A = a_tmp * k["inv_mt"]
        a_sn = a_sn + k["k_ml"] * om * om * a_A
        o_x = lx
        o_v = lv + dt * lx - k["M_fric"] * a_A
        o_th = lth + cs * a_sn - sn * a_cs
        o_om = lom + dt * lth - k["k_jf"] * a_Nn + f32(2.0) * k["k_ml"] * om * sn * a_A
        g_q = k["u_max"] * a_A
        return (np.stack([o_x, o_v, o_th, o_om], axis=1).astype(np.float32),
                g_q.astype(np.float32)[:, None])

    def _mlp_vjp(self, s, q, lam):
        W1, b1, W2, b2, W3, b3 = mlp_unpack(self.weights, self.S + self.C, self.S, self.hidden_sizes)
        _, (xin, h1, h2) = self._mlp_step(s, q)
        d2 = (lam @ W3) * (f32(1.0) - h2 * h2)
        d1 = (d2 @ W2) * (f32(1.0) - h1 * h1)
        din = (d1 @ W1).astype(np.float32)
        return din[:, :self.S].copy(), din[:, self.S:].copy()

    # reference: PredictorWrapper.predict_core(s[N,S], Q[N,H,C]) -> [N,H+1,S]
    # (shape pinned by optimizer_cem_tf.py:70 and Cost_Functions/__init__.py:81)
    def predict_core(self, s: np.ndarray, Q: np.ndarray) -> np.ndarray:
        s = np.asarray(s, dtype=np.float32)
        Q = np.asarray(Q, dtype=np.float32)
        N, H, C = Q.shape
        assert C == self.C and s.shape == (N, self.S)
        traj = np.empty((N, H + 1, self.S), dtype=np.float32)
        traj[:, 0] = s
        cur = s
        if self.kind == "GRU":
            h1 = np.tile(self.hidden[0:1], (N, 1)); h2 = np.tile(self.hidden[1:2], (N, 1))
            for h in range(H):
                cur, h1, h2 = self._gru_step(cur, Q[:, h, :], h1, h2)
                traj[:, h + 1] = cur
            return traj
        for h in range(H):
            cur = self.step(cur, Q[:, h, :])
            traj[:, h + 1] = cur
        return traj


# ----------------------------------------------------------------------------------------------
# Cost (concrete terms build-defined; aggregation follows Cost_Functions/__init__.py)
# ----------------------------------------------------------------------------------------------
class Cost:
    MAX_COST = f32(0.0)   # Cost_Functions/__init__.py:14

    def __init__(self, env, dt: float = 0.02):
        self.env = env
        self.dt = dt
        self.S, self.C = env.S, env.C
        self.quad = isinstance(env, Quad2DParams)
        self.hover = isinstance(env, HoverParams)

    def _k(self):
        return derived_constants(self.env, self.dt, 1)

    # planar quadrotor (build-defined terms: position error, attitude, velocities, input, input change) ------
    def _quad_state_terms(self, states):
        k, e = self._k(), self.env
        dx, dz = states[..., 0] - f32(e.target_x), states[..., 2] - f32(e.target_z)
        pos = k["pos_c"] * (dx * dx + dz * dz)
        ang = f32(e.ang_weight) * (f32(1.0) - np.cos(states[..., 4]))
        return pos.astype(np.float32), ang.astype(np.float32)

    # hovercraft (build-defined terms: position error, attitude, velocities, wheel speed, input, input change) ------------
    def _hover_state_terms(self, states):
        k, e = self._k(), self.env
        dx, dy = states[..., 0] - f32(e.target_x), states[..., 2] - f32(e.target_y)
        pos = k["pos_c"] * (dx * dx + dy * dy)
        ang = f32(e.ang_weight) * (f32(1.0) - np.cos(states[..., 4]))
        return pos.astype(np.float32), ang.astype(np.float32)

    def _hover_stage_cost(self, states, inputs, previous_input):
        k, e = self._k(), self.env
        pos, ang = self._hover_state_terms(states)
        vx, vy, om, w = states[..., 1], states[..., 3], states[..., 5], states[..., 6]
        vel = f32(e.vel_weight) * (vx * vx + vy * vy) + f32(e.angvel_weight) * om * om + f32(e.wheel_weight) * w * w
        du = inputs - self._prev_inputs(inputs, previous_input)
        cc = k["ccR"] * np.sum(inputs * inputs, axis=2, dtype=np.float32)
        ccrc = f32(e.ccrc_weight) * np.sum(du * du, axis=2, dtype=np.float32)
        return (pos + ang + vel + cc + ccrc).astype(np.float32)

    def _prev_inputs(self, inputs, previous_input):
        """[N,H,C]: the input applied one step earlier (previous_input for h = 0)"""
        p0 = np.broadcast_to(np.asarray(previous_input, np.float32).reshape(1, 1, self.C), (inputs.shape[0], 1, self.C))
        return np.concatenate([p0, inputs[:, :-1, :]], axis=1)

    def _quad_stage_cost(self, states, inputs, previous_input):
        k, e = self._k(), self.env
        pos, ang = self._quad_state_terms(states)
        vx, vz, om = states[..., 1], states[..., 3], states[..., 5]
        vel = f32(e.vel_weight) * (vx * vx + vz * vz) + f32(e.angvel_weight) * om * om
        du = inputs - self._prev_inputs(inputs, previous_input)
        cc = k["ccR"] * np.sum(inputs * inputs, axis=2, dtype=np.float32)
        ccrc = f32(e.ccrc_weight) * np.sum(du * du, axis=2, dtype=np.float32)
        return (pos + ang + vel + cc + ccrc).astype(np.float32)

    def _state_terms(self, states):
        k = self._k()
        e = self.env
        x, om, th = states[..., 0], states[..., 3], states[..., 2]
        dxn = (x - f32(e.target_position)) * k["inv_xs"]
        dd = f32(e.dd_weight) * dxn * dxn
        omc = f32(1.0) - np.cos(th)
        ep = k["ep_c"] * omc * omc
        return dd.astype(np.float32), ep.astype(np.float32), om

    def _get_stage_cost(self, states, inputs, previous_input):
        """states [N,H,4], inputs [N,H,1], previous_input [1] -> [N,H] (build-defined terms:
        dd, ep, ekp, cc, ccrc — names from Control_Toolkit_ASF_Template/config_cost_function.yml)."""
        if self.quad:
            return self._quad_stage_cost(states, inputs, previous_input)
        if self.hover:
            return self._hover_stage_cost(states, inputs, previous_input)
        e = self.env
        k = self._k()
        dd, ep, om = self._state_terms(states)
        ekp = f32(e.ekp_weight) * om * om
        u = inputs[..., 0]
        cc = k["ccR"] * u * u
        prev = np.concatenate(
            [np.broadcast_to(np.asarray(previous_input, np.float32).reshape(1, 1), (u.shape[0], 1)),
             u[:, :-1]], axis=1)
        du = u - prev
        ccrc = f32(e.ccrc_weight) * du * du
        return (dd + ep + ekp + cc + ccrc).astype(np.float32)

    def get_stage_cost(self, states, inputs, previous_input):
        # Cost_Functions/__init__.py:63-64
        return self._get_stage_cost(states, inputs, previous_input) - self.MAX_COST

    def get_terminal_cost(self, terminal_states):
        # default in the reference is zeros[N,1] (Cost_Functions/__init__.py:47); the build's
        # concrete cost overrides it with terminal_weight * (dd + ep) (0 by default).
        if self.quad:
            pos, ang = self._quad_state_terms(terminal_states)
            return (f32(self.env.terminal_weight) * (pos + ang)).astype(np.float32)
        if self.hover:
            pos, ang = self._hover_state_terms(terminal_states)
            return (f32(self.env.terminal_weight) * (pos + ang)).astype(np.float32)
        dd, ep, _ = self._state_terms(terminal_states)
        return (f32(self.env.terminal_weight) * (dd + ep)).astype(np.float32)

    def get_summed_stage_cost(self, state_horizon, inputs, previous_input):
        # Cost_Functions/__init__.py:71-72
        return np.sum(self.get_stage_cost(state_horizon[:, :-1, :], inputs, previous_input),
                      axis=1, dtype=np.float32)

    def get_trajectory_cost(self, state_horizon, inputs, previous_input):
        # Cost_Functions/__init__.py:90-93: mean over [H stage costs ‖ 1 terminal cost]
        return aggregate_trajectory_cost(
            self.get_stage_cost(state_horizon[:, :-1, :], inputs, previous_input),
            self.get_terminal_cost(state_horizon[:, -1, :]))

    # gradient pieces for the RPGD adjoint ---------------------------------------------------
    def state_grad(self, states, terminal: bool):
        """d(stage or terminal cost)/d(state) for states [N,S] -> [N,S]."""
        e = self.env
        k = self._k()
        if self.hover:
            z0 = np.zeros_like(states[:, 0])
            gx = f32(2.0) * k["pos_c"] * (states[:, 0] - f32(e.target_x))
            gy = f32(2.0) * k["pos_c"] * (states[:, 2] - f32(e.target_y))
            gth = f32(e.ang_weight) * np.sin(states[:, 4])
            if terminal:
                w = f32(e.terminal_weight)
                return np.stack([w * gx, z0, w * gy, z0, w * gth, z0, z0], 1).astype(np.float32)
            return np.stack([gx, f32(2.0) * f32(e.vel_weight) * states[:, 1], gy, f32(2.0) * f32(e.vel_weight) * states[:, 3], gth,
                             f32(2.0) * f32(e.angvel_weight) * states[:, 5], f32(2.0) * f32(e.wheel_weight) * states[:, 6]], 1).astype(np.float32)
        if self.quad:
            z0 = np.zeros_like(states[:, 0])
            gx = f32(2.0) * k["pos_c"] * (states[:, 0] - f32(e.target_x))
            gz = f32(2.0) * k["pos_c"] * (states[:, 2] - f32(e.target_z))
            gth = f32(e.ang_weight) * np.sin(states[:, 4])
            if terminal:
                w = f32(e.terminal_weight)
                return np.stack([w * gx, z0, w * gz, z0, w * gth, z0], 1).astype(np.float32)
            return np.stack([gx, f32(2.0) * f32(e.vel_weight) * states[:, 1], gz, f32(2.0) * f32(e.vel_weight) * states[:, 3],
                             gth, f32(2.0) * f32(e.angvel_weight) * states[:, 5]], 1).astype(np.float32)
        x, th, om = states[:, 0], states[:, 2], states[:, 3]
        gx = f32(2.0) * f32(e.dd_weight) * k["inv_xs"] * k["inv_xs"] * (x - f32(e.target_position))
        gth = f32(2.0) * k["ep_c"] * (f32(1.0) - np.cos(th)) * np.sin(th)
        if terminal:
            w = f32(e.terminal_weight)
            return np.stack([w * gx, np.zeros_like(gx), w * gth, np.zeros_like(gx)], 1).astype(np.float32)
        gom = f32(2.0) * f32(e.ekp_weight) * om
        return np.stack([gx, np.zeros_like(gx), gth, gom], 1).astype(np.float32)


def input_cost_grad(cost: "Cost", Q, u_prev):
    """d(sum_h stage cost)/dQ through the input-only terms (cc; ccrc towards both neighbours) [N,H,C]; both
    environments share the form ccR*|u|^2 + ccrc_weight*|u - u_prev|^2."""
    k, e = cost._k(), cost.env
    prev = cost._prev_inputs(Q, u_prev)
    gu = f32(2.0) * k["ccR"] * Q + f32(2.0) * f32(e.ccrc_weight) * (Q - prev)
    gu[:, :-1, :] -= f32(2.0) * f32(e.ccrc_weight) * (Q[:, 1:, :] - Q[:, :-1, :])
    return gu.astype(np.float32)


def aggregate_trajectory_cost(stage_costs: np.ndarray, terminal_cost: np.ndarray) -> np.ndarray:
    """Cost_Functions/__init__.py:90-93: concat([stage[N,H], terminal[N,1]], 1).mean(1)."""
    tc = np.asarray(terminal_cost, np.float32).reshape(-1, 1)
    return np.mean(np.concatenate([np.asarray(stage_costs, np.float32), tc], axis=1),
                   axis=1, dtype=np.float32)


# ----------------------------------------------------------------------------------------------
# Interpolator (others/Interpolator.py)
# ----------------------------------------------------------------------------------------------
def num_inducing_points(horizon: int, period: int) -> int:
    # Interpolator.py:79-84
    return int(math.ceil((horizon - 1) / period) + 1)


def interpolation_matrix(horizon: int, period: int, C: int = 1) -> np.ndarray:
    """Interpolator.py:53-77 -> matrix [P, H, C] (after the permute at :76).  Note the quirk kept
    on purpose: the closing row is set to 1 *before* the division by `period` (:73-74), so when
    (H-1) % period == 0 the last horizon step gets weight 1/period, not 1."""
    P = num_inducing_points(horizon, period)
    step = period
    M = np.zeros(((P - 1) * step + 1, P, C), dtype=np.float32)
    blk = np.zeros((step, 2, C), dtype=np.float32)
    for j in range(step):
        blk[j, 0, :] = step - j
        blk[j, 1, :] = j
    for i in range(P - 1):
        M[i * step:(i + 1) * step, i:i + 2, :] = blk
    M[-1, -1, :] = 1
    M = M[:horizon, :, :] / f32(step)
    return np.ascontiguousarray(np.transpose(M, (1, 0, 2))).astype(np.float32)


def interpolate(y: np.ndarray, M: np.ndarray) -> np.ndarray:
    """Interpolator.py:97-106: y[N,P,C] -> [N,H,C]; per control channel y[:, :, c] @ M[:, :, c]."""
    y = np.asarray(y, dtype=np.float32)
    return np.einsum("npc,phc->nhc", y, M).astype(np.float32)


def interpolation_table(horizon: int, period: int):
    """Per-step (i0, w0, w1) such that u[t] = y[i0]*w0 + y[i0+1]*w1 — the two non-zeros of column
    t of the matrix above (what the HIP kernels use).  i0+1 is clamped; its weight is then 0."""
    M = interpolation_matrix(horizon, period, 1)[:, :, 0]  # [P,H]
    P = M.shape[0]
    i0 = np.zeros(horizon, np.int32); w0 = np.zeros(horizon, np.float32); w1 = np.zeros(horizon, np.float32)
    for t in range(horizon):
        nz = np.nonzero(M[:, t])[0]
        assert 1 <= len(nz) <= 2
        i0[t] = min(nz[0], P - 2) if P >= 2 else 0
        if P == 1:
            w0[t] = M[0, t]
        else:
            w0[t] = M[i0[t], t]; w1[t] = M[i0[t] + 1, t]
    return i0, w0, w1


def _limits(low, high, C):
    """control limits as fp32 arrays [C] (the reference's action_low / action_high, Optimizers/__init__.py:42-44)"""
    lo = np.broadcast_to(np.asarray(low, np.float32).reshape(-1), (C,)).astype(np.float32)
    hi = np.broadcast_to(np.asarray(high, np.float32).reshape(-1), (C,)).astype(np.float32)
    if C == 1:
        return f32(lo[0]), f32(hi[0])      # scalars keep the C = 1 arithmetic (and its fixtures) bit-identical
    return lo, hi


def _u_out(x):
    """the optimizer's `u`: fp32 scalar for one control input, [C] array otherwise"""
    x = np.asarray(x, np.float32).reshape(-1)
    return f32(x[0]) if x.size == 1 else x.copy()


# ----------------------------------------------------------------------------------------------
# MPPI (Optimizers/optimizer_mppi.py)
# ----------------------------------------------------------------------------------------------
class MPPI:
    def __init__(self, predictor: Predictor, cost: Cost, low=-1.0, high=1.0, *, num_rollouts, mpc_horizon,
                 cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0, SQRTRHOINV=0.03,
                 period_interpolation_inducing_points=10):
        self.predictor, self.cost = predictor, cost
        self.N, self.H = num_rollouts, mpc_horizon
        self.S, self.C = predictor.S, predictor.C
        self.low, self.high = _limits(low, high, self.C)
        self.cc_weight, self.R, self.LBD, self.NU = f32(cc_weight), f32(R), float(LBD), f32(NU)
        self.period = period_interpolation_inducing_points
        self.P = num_inducing_points(self.H, self.period)
        self.M = interpolation_matrix(self.H, self.period, self.C)
        # optimizer_mppi.py:130 — computed in double, stored as fp32
        self.stdev = f32(np.array(SQRTRHOINV) * (1 / np.sqrt(predictor.dt)))
        self.u = _u_out(np.zeros(self.C, np.float32))
        self.optimizer_reset()

    def optimizer_reset(self):
        # optimizer_mppi.py:227-231
        self.u_nom = (f32(0.5) * (self.low + self.high) * np.ones((1, self.H, self.C), np.float32)).astype(np.float32)

    def mppi_correction_cost(self, u, delta_u):
        # optimizer_mppi.py:154-155 (u = clipped input, delta_u = unclipped perturbation)
        t = f32(0.5) * (f32(1.0) - f32(1.0) / self.NU) * self.R * (delta_u ** 2) + self.R * u * delta_u \
            + f32(0.5) * self.R * (u ** 2)
        return np.sum(self.cc_weight * t, axis=(1, 2), dtype=np.float32)

    def reward_weighted_average(self, S, delta_u):
        # optimizer_mppi.py:163-168
        rho = np.min(S)
        exp_s = np.exp(f32(-1.0 / self.LBD) * (S - rho)).astype(np.float32)
        a = np.sum(exp_s, dtype=np.float32)
        b = np.sum(exp_s[:, None, None] * delta_u, axis=0, dtype=np.float32) / a
        return b.astype(np.float32)

    def step(self, s: np.ndarray, noise: np.ndarray):
        """One MPPI iteration, optimizer_mppi.py:181-193 + :205-225.
        noise = standard-normal draws [N,P,C] (what rng.normal returns at :173-175)."""
        N, H = self.N, self.H
        s = np.asarray(s, np.float32).reshape(1, self.S)
        s_t = np.tile(s, (N, 1))                                                 # :182
        u_nom = np.concatenate([self.u_nom[:, 1:, :], self.u_nom[:, -1:, :]], 1)  # :184
        delta_u = interpolate(np.asarray(noise, np.float32) * self.stdev, self.M)  # :170-179
        u_run = np.clip(np.tile(u_nom, (N, 1, 1)) + delta_u, self.low, self.high)  # :186-187
        traj = self.predictor.predict_core(s_t, u_run)                           # :188
        J = self.cost.get_trajectory_cost(traj, u_run, np.asarray(self.u, np.float32).reshape(self.C)) \
            + self.mppi_correction_cost(u_run, delta_u)                          # :158-161
        u_nom = np.clip(u_nom + self.reward_weighted_average(J, delta_u), self.low, self.high)  # :190
        self.u_nom = u_nom.astype(np.float32)
        self.u = _u_out(u_nom[0, 0, :])                                          # :191
        self.predictor.update(s, self.u)                                         # :192,:195-197 (RNN hidden state)
        self.J, self.u_run, self.rollout_trajectories, self.delta_u = J, u_run, traj, delta_u
        return np.array(self.u, dtype=np.float32)

    def mppi_partials(self, J, noise_scaled):
        """Shard-local (rho_r, a_r, b_r[P]) of SURVEY 8e; merging them reproduces
        reward_weighted_average up to fp32 rounding."""
        rho = np.min(J)
        e = np.exp(f32(-1.0 / self.LBD) * (J - rho)).astype(np.float32)
        return rho, np.sum(e, dtype=np.float32), np.sum(e[:, None, None] * noise_scaled, axis=0, dtype=np.float32)


def merge_mppi_partials(rhos, a_s, b_s, LBD):
    """rho = min rho_r; a = sum a_r e^{-(rho_r-rho)/lbd}; b likewise; return b/a."""
    rhos = np.asarray(rhos, np.float32)
    rho = np.min(rhos)
    sc = np.exp(f32(-1.0 / LBD) * (rhos - rho)).astype(np.float32)
    a = np.sum(np.asarray(a_s, np.float32) * sc, dtype=np.float32)
    b = np.sum(np.asarray(b_s, np.float32) * sc.reshape(-1, *([1] * (np.asarray(b_s).ndim - 1))), axis=0,
               dtype=np.float32)
    return rho, a, (b / a).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# CEM (Optimizers/optimizer_cem_tf.py) — restated from the source text (TF not importable)
# ----------------------------------------------------------------------------------------------
def argsort_total_order(J: np.ndarray) -> np.ndarray:
    """tf.argsort ascending (optimizer_cem_tf.py:73, optimizer_rpgd.py:345) with ties broken by
    index — the total order both the oracle and the HIP kernels fix (SURVEY 7 'hard parts')."""
    return np.argsort(np.asarray(J), kind="stable")


class CEM:
    def __init__(self, predictor, cost, low=-1.0, high=1.0, *, num_rollouts, mpc_horizon, cem_outer_it=3,
                 cem_initial_action_stdev=0.5, cem_stdev_min=0.01, cem_best_k=40, warmup=False,
                 warmup_iterations=250):
        self.predictor, self.cost = predictor, cost
        self.N, self.H = num_rollouts, mpc_horizon
        self.S, self.C = predictor.S, predictor.C
        self.low, self.high = _limits(low, high, self.C)
        self.cem_outer_it, self.K = cem_outer_it, cem_best_k
        self.init_std, self.std_min = f32(cem_initial_action_stdev), f32(cem_stdev_min)
        self.warmup, self.warmup_iterations = warmup, warmup_iterations
        self.optimizer_reset()

    def optimizer_reset(self):
        # optimizer_cem_tf.py:113-117
        self.dist_mue = ((self.low + self.high) * f32(0.5) * np.ones((1, self.H, self.C), np.float32)).astype(np.float32)
        self.stdev = (self.init_std * np.ones((1, self.H, self.C), np.float32)).astype(np.float32)
        self.count = 0
        self.u = _u_out(np.zeros(self.C, np.float32))

    def iterations(self):
        return self.warmup_iterations if (self.warmup and self.count == 0) else self.cem_outer_it  # :92

    def update_distribution(self, s_t, noise):
        # optimizer_cem_tf.py:61-80
        Q = np.clip(np.tile(self.dist_mue, (self.N, 1, 1)) + noise * self.stdev, self.low, self.high)
        traj = self.predictor.predict_core(s_t, Q)
        J = self.cost.get_trajectory_cost(traj, Q, np.asarray(self.u, np.float32).reshape(self.C))
        best = argsort_total_order(J)[: self.K]
        elite = Q[best]
        self.dist_mue = np.mean(elite, axis=0, keepdims=True, dtype=np.float32)
        # tf.math.reduce_std == population std (ddof = 0)
        self.stdev = np.sqrt(np.mean((elite - self.dist_mue) ** 2, axis=0, keepdims=True, dtype=np.float32)).astype(np.float32)
        return Q, elite, J, traj, best

    def step(self, s, noise):
        """noise: standard normal [iterations, N, H, C] (rng.normal at :64-65, one draw per outer it)."""
        s_t = np.tile(np.asarray(s, np.float32).reshape(1, self.S), (self.N, 1))
        its = self.iterations()
        assert noise.shape[0] == its
        for it in range(its):
            Q, elite, J, traj, best = self.update_distribution(s_t, np.asarray(noise[it], np.float32))
        # :99-102
        self.stdev = np.clip(self.stdev, self.std_min, f32(1.0e8))
        self.stdev = np.concatenate([self.stdev[:, 1:, :], self.init_std * np.ones((1, 1, self.C), np.float32)], 1)
        self.u = _u_out(elite[0, 0, :])
        self.dist_mue = np.concatenate(
            [self.dist_mue[:, 1:, :], ((self.low + self.high) * f32(0.5) * np.ones((1, 1, self.C), np.float32)).astype(np.float32)], 1)
        self.Q, self.J, self.rollout_trajectories, self.best_idx = Q, J, traj, best
        self.count += 1
        return np.array(self.u, np.float32)


# ----------------------------------------------------------------------------------------------
# random-action (Optimizers/optimizer_random_action_tf.py) — restated from the source text
# ----------------------------------------------------------------------------------------------
class RandomAction:
    def __init__(self, predictor, cost, low=-1.0, high=1.0, *, num_rollouts, mpc_horizon):
        self.predictor, self.cost = predictor, cost
        self.N, self.H = num_rollouts, mpc_horizon
        self.S, self.C = predictor.S, predictor.C
        self.low, self.high = _limits(low, high, self.C)
        self.u = _u_out(np.zeros(self.C, np.float32))

    def optimizer_reset(self):
        pass  # :78-86 draws and discards a sample; no state

    def step(self, s, u01):
        """u01: U[0,1) draws [N,H,C]; rng.uniform(minval, maxval) = u01*(max-min)+min (:56-61)."""
        s_t = np.tile(np.asarray(s, np.float32).reshape(1, self.S), (self.N, 1))
        Q = (np.asarray(u01, np.float32) * (self.high - self.low) + self.low).astype(np.float32)
        traj = self.predictor.predict_core(s_t, Q)
        J = self.cost.get_trajectory_cost(traj, Q, np.asarray(self.u, np.float32).reshape(self.C))  # :43-45
        best = argsort_total_order(J)[0]                                            # :65-66
        self.u = _u_out(Q[best, 0, :])                                              # :68
        self.Q, self.J, self.rollout_trajectories, self.best_idx = Q, J, traj, best
        return np.array(self.u, np.float32)


# ----------------------------------------------------------------------------------------------
# RPGD (Optimizers/optimizer_rpgd.py), torch branch of ADAM (:56-82)
# ----------------------------------------------------------------------------------------------
def rollout_cost_and_grad(predictor: Predictor, cost: Cost, s_t, Q, u_prev):
    """Forward rollout + hand-written reverse-mode of J.sum() w.r.t. Q (what autograd does at
    optimizer_rpgd.py:310-314 / :329-333).  Returns (J[N], traj[N,H+1,S], dJ/dQ[N,H,C])."""
    N, H, C = Q.shape
    if predictor.kind == "GRU":
        return _rollout_cost_and_grad_gru(predictor, cost, s_t, Q, u_prev)
    traj = predictor.predict_core(s_t, Q)
    J = cost.get_trajectory_cost(traj, Q, u_prev)
    inv = f32(1.0 / (H + 1))   # mean over H+1, Cost_Functions/__init__.py:92
    gu = input_cost_grad(cost, Q, u_prev)   # direct input-cost gradient: cc + ccrc (both neighbours)
    g = np.zeros((N, H, C), np.float32)
    lam = cost.state_grad(traj[:, H], terminal=True) * inv
    for h in range(H - 1, -1, -1):
        ls, gq = predictor.step_vjp(traj[:, h], Q[:, h, :], lam)
        g[:, h, :] = gu[:, h, :] * inv + gq
        lam = cost.state_grad(traj[:, h], terminal=False) * inv + ls
    return J, traj, g.astype(np.float32)


def _rollout_cost_and_grad_gru(predictor: Predictor, cost: Cost, s_t, Q, u_prev):
    """the same for the recurrent predictor: the hidden states of both GRU layers are part of the differentiated path
    (back-propagation through time over the horizon, as autograd does for the reference at optimizer_rpgd.py:329-333);
    every rollout starts from the carried hidden state, which does not depend on Q"""
    N, H, C = Q.shape
    S = predictor.S
    layers, Wo, bo = gru_unpack(predictor.weights, S + C, S)
    h1 = np.tile(predictor.hidden[0:1], (N, 1)); h2 = np.tile(predictor.hidden[1:2], (N, 1))
    traj = np.empty((N, H + 1, S), np.float32)
    traj[:, 0] = s_t
    caches = []
    cur = np.asarray(s_t, np.float32)
    for h in range(H):
        xin = np.concatenate([cur, Q[:, h, :]], axis=1).astype(np.float32)
        h1, c1 = gru_cell_fwd(xin, h1, *layers[0])
        h2, c2 = gru_cell_fwd(h1, h2, *layers[1])
        cur = (h2 @ Wo.T + bo).astype(np.float32)
        traj[:, h + 1] = cur
        caches.append((c1, c2))
    J = cost.get_trajectory_cost(traj, Q, u_prev)
    inv = f32(1.0 / (H + 1))
    gu = input_cost_grad(cost, Q, u_prev)
    g = np.zeros((N, H, C), np.float32)
    lam = cost.state_grad(traj[:, H], terminal=True) * inv
    dh1 = np.zeros((N, GRU_H), np.float32); dh2 = np.zeros((N, GRU_H), np.float32)
    for h in range(H - 1, -1, -1):
        c1, c2 = caches[h]
        dx2, dh2 = gru_cell_bwd((lam @ Wo + dh2).astype(np.float32), c2, layers[1][0], layers[1][1])
        dx1, dh1 = gru_cell_bwd((dx2 + dh1).astype(np.float32), c1, layers[0][0], layers[0][1])
        g[:, h, :] = gu[:, h, :] * inv + dx1[:, S:]
        lam = cost.state_grad(traj[:, h], terminal=False) * inv + dx1[:, :S]
    return J, traj, g.astype(np.float32)


def clip_by_norm(g, clip, axes=(1, 2)):
    """lib.clip_by_norm(g, clip, [1,2]) at optimizer_rpgd.py:315,334 — tf.clip_by_norm semantics:
    g * clip / max(||g||, clip)."""
    nrm = np.sqrt(np.sum(g * g, axis=axes, keepdims=True, dtype=np.float32))
    return (g * f32(clip) / np.maximum(nrm, f32(clip))).astype(np.float32)


class Adam:
    """optimizer_rpgd.py:56-82 (in-repo torch Adam).  Scalars are python doubles rounded to fp32
    when they meet an fp32 tensor, as torch does."""
    def __init__(self, lr, b1, b2, eps):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.reset()

    def reset(self):
        self.step_count, self.m, self.v = 0, None, None   # :133-141

    def apply(self, g, var):
        self.step_count += 1
        if self.m is None:
            self.m, self.v = np.zeros_like(g), np.zeros_like(g)
        self.m = (self.m * f32(self.b1) + f32(1 - self.b1) * g).astype(np.float32)
        self.v = (self.v * f32(self.b2) + f32(1 - self.b2) * (g * g)).astype(np.float32)
        bc1 = f32(1 - self.b1 ** self.step_count)
        bc2 = f32(1 - self.b2 ** self.step_count)
        m_hat, v_hat = self.m / bc1, self.v / bc2
        return (var - f32(self.lr) * m_hat / (np.sqrt(v_hat) + f32(self.eps))).astype(np.float32)


class RPGD:
    def __init__(self, predictor, cost, low=-1.0, high=1.0, *, num_rollouts, mpc_horizon, outer_its=2,
                 sample_stdev=0.5, sample_mean=0.0, sample_whole_control_space=True, uniform_dist_min=-1.0,
                 uniform_dist_max=1.0, resamp_per=10, period_interpolation_inducing_points=10,
                 SAMPLING_DISTRIBUTION="uniform", shift_previous=1, warmup=False, warmup_iterations=250,
                 learning_rate=0.05, opt_keep_k_ratio=0.25, gradmax_clip=5.0, adam_beta_1=0.9,
                 adam_beta_2=0.999, adam_epsilon=1e-8, adam_rule="torch"):
        self.predictor, self.cost = predictor, cost
        self.N, self.H = num_rollouts, mpc_horizon
        self.S, self.C = predictor.S, predictor.C
        self.low, self.high = _limits(low, high, self.C)
        self.outer_its = outer_its
        self.sample_stdev, self.sample_mean = f32(sample_stdev), f32(sample_mean)
        if sample_whole_control_space:                      # optimizer_rpgd.py:200-206
            self.sample_min, self.sample_max = self.low, self.high
        else:
            self.sample_min, self.sample_max = f32(uniform_dist_min), f32(uniform_dist_max)
        self.resamp_per = resamp_per
        self.period = period_interpolation_inducing_points
        self.P = num_inducing_points(self.H, self.period)
        self.M = interpolation_matrix(self.H, self.period, self.C)
        self.dist = SAMPLING_DISTRIBUTION
        self.shift_previous = shift_previous
        self.first_iter_count = warmup_iterations if warmup else outer_its   # :219-221
        self.k = int(max(int(num_rollouts * opt_keep_k_ratio), 1))           # :213
        self.gradmax_clip = f32(gradmax_clip)
        # optimizer_rpgd.py:35-53: the TensorFlow branch wraps tf.keras.optimizers.Adam (third party; published rule, KerasAdam below),
        # the torch branch is the in-repo Adam (:56-82); everything around the update is shared
        if adam_rule not in ("torch", "keras"):
            raise ValueError(f"adam_rule must be 'torch' or 'keras', got {adam_rule!r}")
        self.opt = (KerasAdam if adam_rule == "keras" else Adam)(learning_rate, adam_beta_1, adam_beta_2, adam_epsilon)
        self.u = _u_out(np.zeros(self.C, np.float32))

    def sample_actions(self, draws):
        """optimizer_rpgd.py:275-296.  draws [B,P,C]: standard normal (normal) or U[0,1) (uniform)."""
        d = np.asarray(draws, np.float32)
        if self.dist == "normal":
            Qn = d * self.sample_stdev + self.sample_mean
        elif self.dist == "uniform":
            Qn = d * (self.sample_max - self.sample_min) + self.sample_min   # globals_and_utils.py:83
        else:
            raise ValueError(f"RPGD cannot interpret sampling type {self.dist}")
        Qn = np.clip(Qn, self.low, self.high)
        return interpolate(Qn, self.M)

    def optimizer_reset(self, draws):
        # optimizer_rpgd.py:527-548
        self.Q = self.sample_actions(draws)
        assert self.Q.shape == (self.N, self.H, self.C)
        self.count = 0
        self.opt.reset()
        self.trajectory_ages = np.zeros((self.N,), np.float32)

    def grad_step(self, s_t):
        # optimizer_rpgd.py:329-338
        J, _, g = rollout_cost_and_grad(self.predictor, self.cost, s_t, self.Q, np.asarray(self.u, np.float32).reshape(self.C))
        g = clip_by_norm(g, self.gradmax_clip)
        Qn = self.opt.apply(g, self.Q)
        self.Q = np.clip(Qn, self.low, self.high).astype(np.float32)
        return J

    def step(self, s, resample_draws=None):
        """optimizer_rpgd.py:388-524.  resample_draws [N-k,P,C] is consumed when count % resamp_per == 0."""
        N, k, C = self.N, self.k, self.C
        s_t = np.tile(np.asarray(s, np.float32).reshape(1, self.S), (N, 1))
        iters = self.first_iter_count if self.count == 0 else self.outer_its   # :397-400
        for _ in range(iters):
            self.grad_step(s_t)                                                # :404-406
        # get_action, :340-380
        traj = self.predictor.predict_core(s_t, self.Q)
        J = self.cost.get_trajectory_cost(traj, self.Q, np.asarray(self.u, np.float32).reshape(C))
        best_idx = argsort_total_order(J)[:k]
        sp = self.shift_previous
        Qn = np.concatenate([self.Q[:, sp:, :], np.tile(self.Q[:, -1:, :], (1, sp, 1))], axis=1)
        u_nom = self.Q[None, best_idx[0]]                                      # :426
        m, v = self.opt.m, self.opt.v
        if m is None:   # iters == 0 never happens in practice; keep the oracle total
            m, v = np.zeros_like(self.Q), np.zeros_like(self.Q)
        shift1 = lambda a: np.concatenate([a[:, 1:, :], np.zeros((a.shape[0], 1, C), np.float32)], 1)
        if self.count % self.resamp_per == 0:                                  # :449-495
            Qres = self.sample_actions(resample_draws)
            assert Qres.shape[0] == N - k
            Qn = np.concatenate([Qres, Qn[best_idx]], 0)
            self.trajectory_ages = np.concatenate([np.zeros((N - k,), np.float32), self.trajectory_ages[best_idx]], 0)
            m = np.concatenate([np.zeros((N - k, self.H, C), np.float32), shift1(m[best_idx])], 0)
            v = np.concatenate([np.zeros((N - k, self.H, C), np.float32), shift1(v[best_idx])], 0)
        else:                                                                  # :496-513
            m, v = shift1(m), shift1(v)
        self.opt.m, self.opt.v = m.astype(np.float32), v.astype(np.float32)
        self.trajectory_ages = self.trajectory_ages + f32(1.0)                 # :514
        self.Q_before_warmstart = self.Q
        self.Q = Qn.astype(np.float32)                                         # :515
        self.count += 1
        self.u_nom, self.J, self.rollout_trajectories, self.best_idx = u_nom, J, traj, best_idx
        self.u = _u_out(u_nom[0, 0, :])                                        # :523
        return np.array(self.u, np.float32)


# ----------------------------------------------------------------------------------------------
# Counter-based RNG of the performance mode (Philox4x32-10 + Box-Muller); the device code in
# csrc/ctk_rng.h follows the same published algorithm (Salmon et al., SC'11).  The integer stream
# is bit-exact; the normal transform differs by fp32 rounding of log/sin/cos only.
# ----------------------------------------------------------------------------------------------
_PHILOX_M0, _PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PHILOX_W0, _PHILOX_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32(counter: np.ndarray, key: np.ndarray, rounds: int = 10) -> np.ndarray:
    """counter [...,4] uint32, key [...,2] uint32 -> [...,4] uint32."""
    c = np.array(counter, dtype=np.uint32, copy=True)
    k0 = np.array(key[..., 0], dtype=np.uint32, copy=True)
    k1 = np.array(key[..., 1], dtype=np.uint32, copy=True)
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = _PHILOX_M0 * c0.astype(np.uint64)
            p1 = _PHILOX_M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = (k0 + _PHILOX_W0).astype(np.uint32)
            k1 = (k1 + _PHILOX_W1).astype(np.uint32)
    return np.stack([c0, c1, c2, c3], axis=-1)


def u32_to_unit_open(u: np.ndarray) -> np.ndarray:
    """(0,1]: (u >> 8 + 1) * 2^-24 — never 0, so log() is finite."""
    return ((u >> np.uint32(8)).astype(np.float32) + f32(1.0)) * f32(1.0 / 16777216.0)


def u32_to_unit_halfopen(u: np.ndarray) -> np.ndarray:
    """[0,1): (u >> 8) * 2^-24."""
    return (u >> np.uint32(8)).astype(np.float32) * f32(1.0 / 16777216.0)


def device_noise(seed: int, stream: int, call: int, first_row: int, rows: int, cols: int, kind: str) -> np.ndarray:
    """The draws the device generates for global rows [first_row, first_row+rows), `cols` values
    per row.  Counter = (global_row, block_of_4_cols, call, stream); key = (seed_lo, seed_hi).
    kind: 'normal' (Box-Muller on pairs) or 'uniform' ([0,1))."""
    nblk = (cols + 3) // 4
    r = np.arange(first_row, first_row + rows, dtype=np.uint32)[:, None]
    b = np.arange(nblk, dtype=np.uint32)[None, :]
    ctr = np.stack(np.broadcast_arrays(r, b, np.uint32(call), np.uint32(stream)), axis=-1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    x = philox4x32(ctr, np.broadcast_to(key, ctr.shape[:-1] + (2,)))
    if kind == "uniform":
        out = u32_to_unit_halfopen(x)
    else:
        u1a, u2a = u32_to_unit_open(x[..., 0]), u32_to_unit_halfopen(x[..., 1])
        u1b, u2b = u32_to_unit_open(x[..., 2]), u32_to_unit_halfopen(x[..., 3])
        ra = np.sqrt(f32(-2.0) * np.log(u1a)).astype(np.float32)
        rb = np.sqrt(f32(-2.0) * np.log(u1b)).astype(np.float32)
        ta, tb = f32(2.0 * math.pi) * u2a, f32(2.0 * math.pi) * u2b
        out = np.stack([ra * np.cos(ta), ra * np.sin(ta), rb * np.cos(tb), rb * np.sin(tb)], axis=-1).astype(np.float32)
    return out.reshape(rows, nblk * 4)[:, :cols].astype(np.float32)


# ----------------------------------------------------------------------------------------------
# thin variants.  Restated from the source text (both modules import tensorflow at
# module level and cannot be executed here): parity unpinned by any reference execution.
# ----------------------------------------------------------------------------------------------
class KerasAdam:
    """tf.keras.optimizers.Adam (third party; published update rule, non-amsgrad):
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; var -= lr_t*m/(sqrt(v)+eps)."""
    def __init__(self, lr, b1, b2, eps):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.reset()

    def reset(self):
        self.step_count, self.m, self.v = 0, None, None

    def apply(self, g, var):
        self.step_count += 1
        if self.m is None:
            self.m, self.v = np.zeros_like(g), np.zeros_like(g)
        self.m = (self.m * f32(self.b1) + f32(1 - self.b1) * g).astype(np.float32)
        self.v = (self.v * f32(self.b2) + f32(1 - self.b2) * (g * g)).astype(np.float32)
        lr_t = f32(self.lr) * np.sqrt(f32(1 - self.b2 ** self.step_count)) / f32(1 - self.b1 ** self.step_count)
        return (var - f32(lr_t) * self.m / (np.sqrt(self.v) + f32(self.eps))).astype(np.float32)


class GradientTF:
    """Optimizers/optimizer_gradient_tf.py: Adam descent on N plans, no resampling."""
    def __init__(self, predictor, cost, low=-1.0, high=1.0, *, num_rollouts, mpc_horizon, gradient_steps=5,
                 learning_rate=0.05, adam_beta_1=0.9, adam_beta_2=0.999, adam_epsilon=1e-7, gradmax_clip=5.0,
                 warmup=False, warmup_iterations=250):
        self.predictor, self.cost = predictor, cost
        self.N, self.H = num_rollouts, mpc_horizon
        self.S, self.C = predictor.S, predictor.C
        self.low, self.high = _limits(low, high, self.C)
        self.gradient_steps = gradient_steps
        self.first_iter_count = warmup_iterations if warmup else gradient_steps     # :66-69
        self.gradmax_clip = f32(gradmax_clip)
        self.opt = KerasAdam(learning_rate, adam_beta_1, adam_beta_2, adam_epsilon)
        self.u = _u_out(np.zeros(self.C, np.float32))

    def optimizer_reset(self, u01):
        # :174-185: uniform plans over the whole horizon, Adam weights zeroed
        self.Q = np.clip(np.asarray(u01, np.float32) * (self.high - self.low) + self.low, self.low, self.high).astype(np.float32)
        assert self.Q.shape == (self.N, self.H, self.C)
        self.count = 0
        self.opt.reset()

    def step(self, s, tail_u01):
        """tail_u01 [N,1,1]: the U[0,1) draws behind rng.uniform([N,1,C], low, high) at :137-142."""
        s_t = np.tile(np.asarray(s, np.float32).reshape(1, self.S), (self.N, 1))
        iters = self.first_iter_count if self.count == 0 else self.gradient_steps          # :108-112
        for _ in range(iters):                                                             # :116-118, :82-98
            _, _, g = rollout_cost_and_grad(self.predictor, self.cost, s_t, self.Q, np.asarray(self.u, np.float32).reshape(self.C))
            g = clip_by_norm(g, self.gradmax_clip)
            self.Q = np.clip(self.opt.apply(g, self.Q), self.low, self.high).astype(np.float32)
        traj = self.predictor.predict_core(s_t, self.Q)                                    # :127
        J = self.cost.get_trajectory_cost(traj, self.Q, np.asarray(self.u, np.float32).reshape(self.C))
        best = argsort_total_order(J)[0]                                                   # :130-131
        self.u = _u_out(self.Q[best, 0, :])                                                   # :133
        self.J, self.Q_refined, self.best_idx = J, self.Q.copy(), best
        self.count += 1
        Q_s = (np.asarray(tail_u01, np.float32).reshape(self.N, 1, self.C) * (self.high - self.low) + self.low).astype(np.float32)
        self.Q = np.concatenate([self.Q[:, 1:, :], Q_s], axis=1)                            # :143-144
        if self.opt.m is not None:                                                         # :146-166
            z = np.zeros((self.N, 1, self.C), np.float32)
            self.opt.m = np.concatenate([self.opt.m[:, 1:, :], z], 1)
            self.opt.v = np.concatenate([self.opt.v[:, 1:, :], z], 1)
        return np.array(self.u, np.float32)


class CEMNaiveGrad:
    """Optimizers/optimizer_cem_naive_grad_tf.py: CEM whose samples take one clipped-gradient SGD step."""
    def __init__(self, predictor, cost, low=-1.0, high=1.0, *, num_rollouts, mpc_horizon, cem_outer_it=1,
                 cem_initial_action_stdev=0.5, cem_stdev_min=0.1, cem_best_k=40, learning_rate=0.1, gradmax_clip=10.0):
        self.predictor, self.cost = predictor, cost
        self.N, self.H = num_rollouts, mpc_horizon
        self.S, self.C = predictor.S, predictor.C
        self.low, self.high = _limits(low, high, self.C)
        self.cem_outer_it, self.K = cem_outer_it, cem_best_k
        self.init_std, self.std_min = f32(cem_initial_action_stdev), f32(cem_stdev_min)
        self.lr, self.gradmax_clip = f32(learning_rate), f32(gradmax_clip)
        self.u = _u_out(np.zeros(self.C, np.float32))
        self.optimizer_reset()

    def optimizer_reset(self):
        # :117-119
        self.dist_mue = ((self.low + self.high) * f32(0.5) * np.ones((1, self.H, self.C), np.float32)).astype(np.float32)
        self.stdev = (self.init_std * np.ones((1, self.H, self.C), np.float32)).astype(np.float32)

    def step(self, s, noise):
        """noise: standard normal [cem_outer_it, N, H, 1] (rng.normal at :60-61, one draw per outer iteration)."""
        s_t = np.tile(np.asarray(s, np.float32).reshape(1, self.S), (self.N, 1))
        up = np.asarray(self.u, np.float32).reshape(self.C)
        for it in range(self.cem_outer_it):                                                # :96-97
            Q = np.clip(np.tile(self.dist_mue, (self.N, 1, 1)) + np.asarray(noise[it], np.float32) * self.stdev,
                        self.low, self.high).astype(np.float32)                           # :60-62
            _, _, g = rollout_cost_and_grad(self.predictor, self.cost, s_t, Q, up)         # :64-68
            Qn = np.clip(Q - self.lr * clip_by_norm(g, self.gradmax_clip), self.low, self.high).astype(np.float32)   # :70-72
            traj = self.predictor.predict_core(s_t, Qn)                                    # :74-75
            J = self.cost.get_trajectory_cost(traj, Qn, up)
            best = argsort_total_order(J)[: self.K]                                        # :78-80
            elite = Qn[best]
            self.dist_mue = np.mean(elite, axis=0, keepdims=True, dtype=np.float32)        # :82-83
            self.stdev = np.sqrt(np.mean((elite - self.dist_mue) ** 2, axis=0, keepdims=True, dtype=np.float32)).astype(np.float32)
        self.stdev = np.clip(self.stdev, self.std_min, f32(10.0))                          # :101
        self.stdev = np.concatenate([self.stdev[:, 1:, :], self.init_std * np.ones((1, 1, self.C), np.float32)], 1)   # :102
        self.u = _u_out(self.dist_mue[0, 0, :])                                               # :103 — the MEAN's first input
        self.dist_mue = np.concatenate(
            [self.dist_mue[:, 1:, :], (self.low + self.high) * f32(0.5) * np.ones((1, 1, self.C), np.float32)], 1)   # :104
        self.Q, self.J, self.best_idx = Qn, J, best
        return np.array(self.u, np.float32)


class CEMGradBharadhwaj:
    """Optimizers/optimizer_cem_grad_bharadhwaj_tf.py: CEM over [elites | fresh samples] with one Keras-Adam
    step per outer iteration.  The Keras optimizer's moments live by POSITION in the population variable and
    are never shifted or reset by the reference (optimizer_reset :180-184 resets only the distribution)."""
    def __init__(self, predictor, cost, low=-1.0, high=1.0, *, num_rollouts, mpc_horizon, cem_outer_it=2,
                 cem_initial_action_stdev=2.0, cem_stdev_min=1e-6, cem_best_k=8, learning_rate=0.05, adam_beta_1=0.9,
                 adam_beta_2=0.999, adam_epsilon=1e-8, gradmax_clip=5.0, warmup=False, warmup_iterations=250):
        self.predictor, self.cost = predictor, cost
        self.N, self.H = num_rollouts, mpc_horizon
        self.S, self.C = predictor.S, predictor.C
        self.low, self.high = _limits(low, high, self.C)
        self.cem_outer_it, self.K = cem_outer_it, cem_best_k
        self.init_std, self.std_min = f32(cem_initial_action_stdev), f32(cem_stdev_min)
        self.gradmax_clip = f32(gradmax_clip)
        self.warmup, self.warmup_iterations = warmup, warmup_iterations
        self.opt = KerasAdam(learning_rate, adam_beta_1, adam_beta_2, adam_epsilon)
        self.u = _u_out(np.zeros(self.C, np.float32))
        self.optimizer_reset()

    def optimizer_reset(self):
        self.dist_mue = ((self.low + self.high) * f32(0.5) * np.ones((1, self.H, self.C), np.float32)).astype(np.float32)
        self.stdev = (self.init_std * np.ones((1, self.H, self.C), np.float32)).astype(np.float32)
        self.count = 0

    def iterations(self):
        return self.warmup_iterations if (self.warmup and self.count == 0) else self.cem_outer_it   # :161

    def _sample(self, eps):
        return (np.tile(self.dist_mue, (eps.shape[0], 1, 1)) + self.stdev * np.asarray(eps, np.float32)).astype(np.float32)   # :122-128

    def step(self, s, eps_elite, eps_rest):
        """eps_elite [K,H,1] (:158), eps_rest [iterations, N-K, H, 1] (:94): standard normal draws."""
        s_t = np.tile(np.asarray(s, np.float32).reshape(1, self.S), (self.N, 1))
        up = np.asarray(self.u, np.float32).reshape(self.C)
        elite_Q = self._sample(eps_elite)                                                   # :158
        for it in range(self.iterations()):                                                 # :162-163
            Q = np.clip(np.concatenate([elite_Q, self._sample(eps_rest[it])], 0), self.low, self.high).astype(np.float32)   # :94-96
            _, _, g = rollout_cost_and_grad(self.predictor, self.cost, s_t, Q, up)          # :99-104
            g = clip_by_norm(g, self.gradmax_clip)                                          # :106
            Qn = np.clip(self.opt.apply(g, Q), self.low, self.high).astype(np.float32)      # :108-109
            traj = self.predictor.predict_core(s_t, Qn)                                     # :111-112
            J = self.cost.get_trajectory_cost(traj, Qn, up)
            best = argsort_total_order(J)[: self.K]                                         # :115-117
            elite_Q = Qn[best]
            self.dist_mue = np.mean(elite_Q, axis=0, keepdims=True, dtype=np.float32)       # :119-120
            self.stdev = np.sqrt(np.mean((elite_Q - self.dist_mue) ** 2, axis=0, keepdims=True, dtype=np.float32)).astype(np.float32)
        self.u = _u_out(elite_Q[0, 0, :])                                                      # :167
        # apply_time_delta :130-141
        self.dist_mue = np.concatenate(
            [self.dist_mue[:, 1:, :], (self.low + self.high) * f32(0.5) * np.ones((1, 1, self.C), np.float32)], 1)
        self.stdev = np.clip(self.stdev, self.std_min, f32(10.0))
        self.stdev = np.concatenate([self.stdev[:, 1:, :], self.init_std * np.ones((1, 1, self.C), np.float32)], 1)
        self.Q, self.J, self.best_idx = Qn, J, best
        self.count += 1
        return np.array(self.u, np.float32)
