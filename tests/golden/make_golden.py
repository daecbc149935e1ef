#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by EXECUTING the unmodified reference
modules (read from /root/reference, build container only) on torch-CPU.

The reference's third-party imports (SI_Toolkit, watchdog, Control_Toolkit_ASF) are not vendored
with it; `tests/golden/standins/` supplies build-authored, test-only stand-ins (see its README).
What the fixtures therefore pin is the logic that IS in the reference:
  others/Interpolator.py, Cost_Functions/__init__.py (aggregation), Optimizers/optimizer_mppi.py,
  Optimizers/optimizer_rpgd.py (ADAM torch branch + RPGD step), others/globals_and_utils.py
  (create_rng / torch_gen_like_TF), Controllers/controller_mpc.py + Controllers/__init__.py
  (construction order and step plumbing).
Round 4: Optimizers/optimizer_cem_tf.py, optimizer_random_action_tf.py and optimizer_cem_naive_grad_tf.py too — their module-level
`import tensorflow as tf` resolves to a torch-backed, build-authored stand-in (standins/tensorflow; its semantic choices — stable
ascending argsort, population std — are listed in standins/README.md), `computation_library: tensorflow` in the work
directory's config_controllers.yml, driven through the reference's own controller_mpc like MPPI.
What they do NOT pin (build-defined, "parity unpinned"): the predictor, the concrete cost terms, the meaning of the `lib.*` / `tf.*`
primitives (the stand-ins'), and everything that needs tf.keras.optimizers.Adam (optimizer_gradient_tf, cem_grad_bharadhwaj, RPGD's
TF branch).

Usage (build container only):  python tests/golden/make_golden.py
Neither this script nor the stand-ins run on the GPU box; only the .npz outputs travel.
"""
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path.insert(0, REPO)

from oracle import ctk_oracle as O  # noqa: E402  (only for EnvParams/derived constants + plant step)


class RecordingRng:
    """Observer around the reference's generator (globals_and_utils.py:61-83): calls it exactly as
    the optimizer does and additionally recovers the raw N(0,1) / U[0,1) draws from a snapshot of
    the generator state, asserting that the reference's scaling of them gives its output."""

    def __init__(self, gen):
        self.gen = gen
        self.raw = []

    def _snapshot(self):
        g = torch.Generator()
        g.set_state(self.gen.rng.get_state())
        return g

    def normal(self, shape, dtype, mean=0.0, stddev=1.0):
        g = self._snapshot()
        out = self.gen.normal(shape, dtype=dtype, mean=mean, stddev=stddev)
        raw = torch.normal(mean=0.0, std=1.0, size=shape, generator=g, dtype=dtype)
        assert torch.equal(raw * stddev + mean, out) or torch.allclose(raw * stddev + mean, out, rtol=0, atol=0)
        self.raw.append(raw.numpy().copy())
        return out

    def uniform(self, shape, dtype, minval=0.0, maxval=1.0):
        g = self._snapshot()
        out = self.gen.uniform(shape, dtype=dtype, minval=minval, maxval=maxval)
        raw = torch.rand(*shape, generator=g, dtype=dtype)
        assert torch.equal(raw * (maxval - minval) + minval, out)
        self.raw.append(raw.numpy().copy())
        return out


def setup_workdir():
    work = tempfile.mkdtemp(prefix="ctk_golden_")
    shutil.copytree(os.path.join(HERE, "standins", "asf_template", "Control_Toolkit_ASF"),
                    os.path.join(work, "Control_Toolkit_ASF"))
    os.symlink(REFERENCE, os.path.join(work, "Control_Toolkit"))
    os.chdir(work)
    sys.path.insert(0, work)
    sys.path.insert(0, os.path.join(HERE, "standins"))
    sys.dont_write_bytecode = True
    return work


def inject_constants(env: O.EnvParams, dt: float, mlp_weights):
    import SI_Toolkit.Predictors.predictor_wrapper as pw
    import Control_Toolkit_ASF.Cost_Functions.CartPole.default as cf
    k = {kk: float(v) for kk, v in O.derived_constants(env, dt, 1).items()}
    # fp32 scalars as python floats: torch multiplies them in fp32 with fp32 tensors
    pw.CONSTANTS.clear(); pw.CONSTANTS.update(k)
    pw.MLP_WEIGHTS = tuple(torch.tensor(a) for a in O.mlp_unpack(mlp_weights))
    cf.CONSTANTS.clear(); cf.CONSTANTS.update(k)
    for name in ("target_position", "dd_weight", "ekp_weight", "ccrc_weight", "terminal_weight"):
        cf.CONSTANTS[name] = float(np.float32(getattr(env, name)))


def inject_quad(env: O.Quad2DParams, dt: float):
    """constants of the second environment (planar quadrotor, 6 states / 2 inputs) into its torch stand-ins"""
    import SI_Toolkit.Predictors.predictor_wrapper as pw
    import Control_Toolkit_ASF.Cost_Functions.Quad2D.default as cf
    k = {kk: float(v) for kk, v in O.quad2d_constants(env, dt, 1).items()}
    pw.CONSTANTS.clear(); pw.CONSTANTS.update(k)
    cf.CONSTANTS.clear(); cf.CONSTANTS.update(k)
    for name in ("target_x", "target_z", "ang_weight", "vel_weight", "angvel_weight", "ccrc_weight", "terminal_weight"):
        cf.CONSTANTS[name] = float(np.float32(getattr(env, name)))


def inject_hover(env: O.HoverParams, dt: float, mlp_weights):
    """constants of the third environment (hovercraft with a reaction wheel, 7 states / 3 inputs) and its 10-32-32-7 MLP"""
    import SI_Toolkit.Predictors.predictor_wrapper as pw
    import Control_Toolkit_ASF.Cost_Functions.Hover.default as cf
    k = {kk: float(v) for kk, v in O.hover_constants(env, dt, 1).items()}
    pw.CONSTANTS.clear(); pw.CONSTANTS.update(k)
    pw.MLP_WEIGHTS = tuple(torch.tensor(a) for a in O.mlp_unpack(mlp_weights, 10, 7))
    cf.CONSTANTS.clear(); cf.CONSTANTS.update(k)
    for name in ("target_x", "target_y", "ang_weight", "vel_weight", "angvel_weight", "wheel_weight", "ccrc_weight", "terminal_weight"):
        cf.CONSTANTS[name] = float(np.float32(getattr(env, name)))


def plant_step(pred: O.Predictor, s, u):
    return pred.step(np.asarray(s, np.float32).reshape(1, pred.S), np.asarray(u, np.float32).reshape(1, pred.C))[0]


def initial_state(seed):
    rng = np.random.default_rng(seed)   # SURVEY 8d common synthetic inputs
    return np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.5, 0.5), rng.uniform(-np.pi, np.pi),
                     rng.uniform(-2, 2)], dtype=np.float32)


def set_computation_library(name):
    """`computation_library:` of the work directory's config_controllers.yml, which template_controller.__init__ reads at every
    construction (Controllers/__init__.py:40-58)"""
    import yaml
    path = os.path.join("Control_Toolkit_ASF", "config_controllers.yml")
    with open(path) as f:
        cfg = yaml.safe_load(f)
    cfg["mpc"]["computation_library"] = name
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)


def set_controller_key(key, value):
    """one key of the work directory's config_controllers.yml (read by template_controller.__init__ at every construction)"""
    import yaml
    path = os.path.join("Control_Toolkit_ASF", "config_controllers.yml")
    with open(path) as f:
        cfg = yaml.safe_load(f)
    cfg["mpc"][key] = value
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)


def record_rpgd_logged_outputs(d, build_controller, steps):
    """Second pass over an RPGD case with `controller_logging` and `calculate_optimal_trajectory` on (same seed => same draws): what
    optimizer_rpgd.py:428-433 puts into logging_values — the DESCENDED population, its costs and get_action's rollout_trajectories
    (:424), the ages as logged (before the step's update, :432) — and :518-521's optimal_trajectory / summed_stage_cost."""
    set_controller_key("calculate_optimal_trajectory", True)
    try:
        ctrl = build_controller()
    finally:
        set_controller_key("calculate_optimal_trajectory", False)
    # optimizer_logging only: with controller_logging the reference's update_logs (Controllers/__init__.py:176-178) calls .copy() on
    # `u_logged`, which RPGD fills with self.u BEFORE updating it (:433 vs :523) — the float 0.0 of Optimizers/__init__.py:34 on step 0
    opt = ctrl.optimizer
    opt.optimizer_logging = True
    assert opt.calculate_optimal_trajectory and not ctrl.controller_logging
    for t in range(steps):
        u = ctrl.step(d[f"s_{t}"].copy())
        assert np.array_equal(np.asarray(u, np.float32).reshape(-1), d[f"u_{t}"])
        lv = opt.logging_values
        d[f"Q_logged_{t}"] = np.asarray(lv["Q_logged"]).copy()
        d[f"J_logged_{t}"] = np.asarray(lv["J_logged"]).copy()
        d[f"traj_logged_{t}"] = np.asarray(lv["rollout_trajectories_logged"]).copy()
        d[f"ages_logged_{t}"] = np.asarray(lv["trajectory_ages_logged"]).copy()
        d[f"optimal_trajectory_{t}"] = np.asarray(opt.optimal_trajectory).copy()
        d[f"summed_stage_cost_{t}"] = np.asarray(opt.summed_stage_cost, np.float32).reshape(-1).copy()


def record_tf_only_optimizers(out_dir, cm, envs, dt):
    """optimizer_cem_tf / optimizer_random_action_tf / optimizer_cem_naive_grad_tf, UNMODIFIED, driven through the reference's own
    controller_mpc with `computation_library: tensorflow`.  Their `import tensorflow as tf` resolves to the torch-backed stand-in
    (standins/tensorflow: semantic choices listed there and in standins/README.md), `TensorFlowLibrary` to the SI_Toolkit stand-in's."""
    import Control_Toolkit.Optimizers as tmpl
    import SI_Toolkit.Predictors.predictor_wrapper as pw

    def controller(opt_name, cfg, pred, envname):
        e = envs[envname]
        e["inject"]()
        pw.ENVIRONMENT = envname
        orig_create, holder = tmpl.create_rng, {}

        def recording_create(id, seed, computation_library=None):
            holder["rec"] = RecordingRng(orig_create(id, seed, computation_library=computation_library))
            return holder["rec"]
        tmpl.create_rng = recording_create
        try:
            cm.config_optimizers[opt_name] = dict(cfg)
            ctrl = cm.controller_mpc(envname, (e["low"], e["high"]), {})
            ctrl.controller_logging = True
            ctrl.configure(optimizer_name=opt_name, predictor_specification=pred)
        finally:
            tmpl.create_rng = orig_create
        assert ctrl.lib.lib == "TF" and ctrl.optimizer.optimizer_logging
        return ctrl, holder["rec"]

    def header(c, cfg, envname):
        e = envs[envname]
        d = dict(e["common"], low=e["low"], high=e["high"], predictor=np.array(c["pred"]), environment=np.array(envname),
                 **{k: (np.float32(v) if isinstance(v, float) else np.array(v)) for k, v in cfg.items()})
        if c["pred"] != "MLP":
            d.pop("mlp_weights", None)
        return d

    def u_prev_of(opt, C):
        return np.broadcast_to(np.asarray(opt.u, np.float32).reshape(-1), (C,)).copy()

    set_computation_library("tensorflow")
    try:
        # ---- CEM (Optimizers/optimizer_cem_tf.py:54-117) --------------------------------------------------------------
        cem_cases = {
            "tiny":    dict(env="CartPole", pred="ODE", N=16, H=8, K=4, its=2, steps=3, seed=41, traj=True),
            "default": dict(env="CartPole", pred="ODE", N=200, H=40, K=40, its=3, steps=3, seed=42, traj=True),   # config_optimizers.yml:5-14
            "cfg3":    dict(env="CartPole", pred="ODE", N=4096, H=30, K=409, its=3, steps=3, seed=43, traj=False),  # BASELINE configs[2]
            "warmup":  dict(env="CartPole", pred="ODE", N=64, H=12, K=8, its=2, steps=3, seed=44, traj=True, warmup=True, warmup_iterations=5),
            "mlp":     dict(env="CartPole", pred="MLP", N=128, H=20, K=16, its=2, steps=3, seed=45, traj=True),
            "quad2d":  dict(env="Quad2D", pred="ODE", N=128, H=20, K=20, its=3, steps=3, seed=46, traj=True),
            "hover":   dict(env="Hover", pred="ODE", N=96, H=16, K=12, its=2, steps=3, seed=47, traj=True),
            "hover_mlp": dict(env="Hover", pred="MLP", N=64, H=12, K=9, its=3, steps=3, seed=48, traj=True),
        }
        for name, c in cem_cases.items():
            e = envs[c["env"]]
            cfg = dict(seed=1, mpc_horizon=c["H"], cem_outer_it=c["its"], cem_initial_action_stdev=0.5, num_rollouts=c["N"],
                       cem_stdev_min=0.01, cem_best_k=c["K"], warmup=c.get("warmup", False),
                       warmup_iterations=c.get("warmup_iterations", 250), mpc_timestep=dt)
            ctrl, rec = controller("cem-tf", cfg, c["pred"], c["env"])
            opt = ctrl.optimizer
            d = header(c, cfg, c["env"])
            d["dist_mue_init"] = opt.dist_mue.numpy().copy(); d["stdev_init"] = opt.stdev.numpy().copy()
            plant = O.Predictor(kind="ODE", dt=dt, env=e["env"])
            s = e["state"](c["seed"])
            for t in range(c["steps"]):
                ndraw = len(rec.raw)
                u_prev = u_prev_of(opt, e["C"])
                u = ctrl.step(s.copy())
                lv = opt.logging_values
                d[f"s_{t}"] = s.copy(); d[f"u_prev_{t}"] = u_prev
                d[f"noise_{t}"] = np.stack(rec.raw[ndraw:])                     # [iterations, N, H, C]: one rng.normal per outer iteration (:64-65)
                d[f"u_{t}"] = np.asarray(u, np.float32).reshape(-1)
                d[f"dist_mue_{t}"] = opt.dist_mue.numpy().copy(); d[f"stdev_{t}"] = opt.stdev.numpy().copy()
                d[f"J_{t}"] = np.asarray(lv["J_logged"]).copy()                 # last iteration's costs / plans (:96,:105-107)
                d[f"Q_{t}"] = np.asarray(lv["Q_logged"]).copy()
                if c["traj"]:
                    d[f"traj_{t}"] = np.asarray(lv["rollout_trajectories_logged"]).copy()
                s = plant_step(plant, s, u)
            assert opt.count == c["steps"]
            d["steps"] = np.int32(c["steps"])
            np.savez_compressed(os.path.join(out_dir, f"cem_{name}.npz"), **d)

        # ---- random-action (Optimizers/optimizer_random_action_tf.py:38-86) -------------------------------------------
        random_cases = {
            "cfg1":    dict(env="CartPole", pred="ODE", N=32, H=10, steps=3, seed=51),     # BASELINE configs[0]
            "default": dict(env="CartPole", pred="ODE", N=320, H=35, steps=3, seed=52),    # config_optimizers.yml:212-215
            "quad2d":  dict(env="Quad2D", pred="ODE", N=64, H=12, steps=3, seed=53),
            "hover_mlp": dict(env="Hover", pred="MLP", N=48, H=9, steps=2, seed=54),
        }
        for name, c in random_cases.items():
            e = envs[c["env"]]
            cfg = dict(seed=1, mpc_horizon=c["H"], num_rollouts=c["N"], mpc_timestep=dt)
            ctrl, rec = controller("random-action-tf", cfg, c["pred"], c["env"])
            opt = ctrl.optimizer
            assert len(rec.raw) == 1            # optimizer_reset draws one population and drops it (:78-86): it advances the stream
            d = header(c, cfg, c["env"])
            plant = O.Predictor(kind="ODE", dt=dt, env=e["env"])
            s = e["state"](c["seed"])
            for t in range(c["steps"]):
                u_prev = u_prev_of(opt, e["C"])
                u = ctrl.step(s.copy())
                lv = opt.logging_values
                d[f"s_{t}"] = s.copy(); d[f"u_prev_{t}"] = u_prev
                d[f"u01_{t}"] = rec.raw[-1]
                d[f"u_{t}"] = np.asarray(u, np.float32).reshape(-1)
                d[f"Q_{t}"] = np.asarray(lv["Q_logged"]).copy(); d[f"J_{t}"] = np.asarray(lv["J_logged"]).copy()
                d[f"traj_{t}"] = np.asarray(lv["rollout_trajectories_logged"]).copy()
                s = plant_step(plant, s, u)
            d["steps"] = np.int32(c["steps"])
            np.savez_compressed(os.path.join(out_dir, f"random_{name}.npz"), **d)

        # ---- CEM + one clipped-gradient step (Optimizers/optimizer_cem_naive_grad_tf.py:58-119) -----------------------
        naive_cases = {
            "default": dict(env="CartPole", pred="ODE", N=200, H=35, K=40, its=1, steps=3, seed=61),   # config_optimizers.yml:23-32
            "its2":    dict(env="CartPole", pred="MLP", N=64, H=20, K=10, its=2, steps=3, seed=62),
            "hover":   dict(env="Hover", pred="ODE", N=48, H=12, K=8, its=2, steps=3, seed=63),
        }
        for name, c in naive_cases.items():
            e = envs[c["env"]]
            cfg = dict(seed=1, mpc_horizon=c["H"], cem_outer_it=c["its"], num_rollouts=c["N"], cem_stdev_min=0.1,
                       cem_initial_action_stdev=0.5, cem_best_k=c["K"], learning_rate=0.1, gradmax_clip=10.0, mpc_timestep=dt)
            ctrl, rec = controller("cem-naive-grad-tf", cfg, c["pred"], c["env"])
            opt = ctrl.optimizer
            d = header(c, cfg, c["env"])
            plant = O.Predictor(kind="ODE", dt=dt, env=e["env"])
            s = e["state"](c["seed"])
            for t in range(c["steps"]):
                ndraw = len(rec.raw)
                u_prev = u_prev_of(opt, e["C"])
                u = ctrl.step(s.copy())
                lv = opt.logging_values
                d[f"s_{t}"] = s.copy(); d[f"u_prev_{t}"] = u_prev
                d[f"noise_{t}"] = np.stack(rec.raw[ndraw:])
                d[f"u_{t}"] = np.asarray(u, np.float32).reshape(-1)
                d[f"dist_mue_{t}"] = opt.dist_mue.numpy().copy(); d[f"stdev_{t}"] = opt.stdev.numpy().copy()
                d[f"J_{t}"] = np.asarray(lv["J_logged"]).copy(); d[f"Q_{t}"] = np.asarray(lv["Q_logged"]).copy()
                s = plant_step(plant, s, u)
            d["steps"] = np.int32(c["steps"])
            np.savez_compressed(os.path.join(out_dir, f"cem_naive_grad_{name}.npz"), **d)
    finally:
        set_computation_library("pytorch")
        pw.ENVIRONMENT = "CartPole"


def main():
    out_dir = HERE
    setup_workdir()
    env = O.EnvParams(terminal_weight=0.5)
    dt = 0.02
    mlp_w = O.mlp_default_weights(0)
    inject_constants(env, dt, mlp_w)

    from SI_Toolkit.computation_library import PyTorchLibrary
    from Control_Toolkit.others.Interpolator import Interpolator
    from Control_Toolkit.others.globals_and_utils import create_rng
    import Control_Toolkit.Controllers.controller_mpc as cm
    from Control_Toolkit.Optimizers.optimizer_rpgd import ADAM
    from Control_Toolkit_ASF.Cost_Functions.CartPole.default import default as CostClass

    lib = PyTorchLibrary()
    env_arr = env.as_array()
    common = dict(env_params=env_arr, env_param_names=np.array(O.PARAM_NAMES), dt=np.float32(dt),
                  mlp_weights=mlp_w)

    # ---- Interpolator (others/Interpolator.py) ------------------------------------------------
    interp = {}
    g = np.random.default_rng(11)
    for (H, p) in [(10, 1), (50, 10), (50, 1), (30, 10), (100, 10), (43, 10), (41, 10), (5, 10), (12, 3)]:
        for C in (1, 2):
            I = Interpolator(H, p, C, lib)
            P = I.number_of_interpolation_inducing_points
            y = g.standard_normal((7, P, C)).astype(np.float32)
            out = I.interpolate(torch.tensor(y)).numpy()
            interp[f"H{H}_p{p}_C{C}_y"] = y
            interp[f"H{H}_p{p}_C{C}_out"] = out
            interp[f"H{H}_p{p}_C{C}_mat"] = I.interp_mat.numpy()   # [P,H,C]
    np.savez_compressed(os.path.join(out_dir, "interpolator.npz"), **interp)

    # ---- cost aggregation (Cost_Functions/__init__.py:38-93) ----------------------------------
    class _Agg(CostClass.__mro__[1]):   # reference cost_function_base with canned stage/terminal costs
        def __init__(self, stage, term):
            self.lib, self._s, self._t = lib, stage, term
        def _get_stage_cost(self, states, inputs, previous_input):
            return self._s
        def get_terminal_cost(self, terminal_states):
            return self._t
    stage = g.standard_normal((9, 13)).astype(np.float32) * 100
    term = g.standard_normal((9,)).astype(np.float32) * 100
    agg = _Agg(torch.tensor(stage), torch.tensor(term))
    J = agg.get_trajectory_cost(torch.zeros(9, 14, 4), torch.zeros(9, 13, 1), torch.zeros(1)).numpy()
    Jsum = agg.get_summed_stage_cost(torch.zeros(9, 14, 4), torch.zeros(9, 13, 1), torch.zeros(1)).numpy()
    # full concrete cost through the reference base class
    from types import SimpleNamespace
    cc = CostClass(SimpleNamespace(), lib)
    traj = g.standard_normal((6, 8, 4)).astype(np.float32)
    inp = g.uniform(-1, 1, (6, 7, 1)).astype(np.float32)
    Jfull = cc.get_trajectory_cost(torch.tensor(traj), torch.tensor(inp), np.float32(0.25)).numpy()
    np.savez_compressed(os.path.join(out_dir, "cost_aggregation.npz"), stage=stage, terminal=term, J=J, J_summed=Jsum,
                        traj=traj, inputs=inp, u_prev=np.float32(0.25), J_full=Jfull, **common)

    # ---- controller_mpc helper ----------------------------------------------------------------
    low, high = np.array([-1.0], np.float32), np.array([1.0], np.float32)

    def make_controller(opt_name, opt_cfg, predictor_spec, environment="CartPole", limits=None):
        cm.config_optimizers[opt_name] = dict(opt_cfg)
        ctrl = cm.controller_mpc(environment, limits if limits is not None else (low, high), {})
        ctrl.configure(optimizer_name=opt_name, predictor_specification=predictor_spec)
        return ctrl

    # ---- MPPI (Optimizers/optimizer_mppi.py via Controllers/controller_mpc.py) -----------------
    mppi_cases = {
        "tiny_ode":   dict(N=8, H=10, p=1, pred="ODE", steps=3, seed=0, keep_traj=True),
        "interp_ode": dict(N=64, H=50, p=10, pred="ODE", steps=3, seed=1, keep_traj=True),
        "cfg2_ode":   dict(N=1024, H=50, p=1, pred="ODE", steps=3, seed=2, keep_traj=False),
        "quirk_ode":  dict(N=32, H=41, p=10, pred="ODE", steps=2, seed=3, keep_traj=True),
        "mlp":        dict(N=64, H=30, p=7, pred="MLP", steps=3, seed=4, keep_traj=True),
    }
    for name, c in mppi_cases.items():
        cfg = dict(seed=1, mpc_horizon=c["H"], num_rollouts=c["N"], cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0,
                   SQRTRHOINV=0.03, period_interpolation_inducing_points=c["p"], mpc_timestep=dt)
        ctrl = make_controller("mppi", cfg, c["pred"])
        opt = ctrl.optimizer
        rec = RecordingRng(opt.rng); opt.rng = rec
        plant = O.Predictor(kind="ODE", dt=dt, env=env)
        s = initial_state(c["seed"])
        d = dict(common, low=low, high=high, predictor=np.array(c["pred"]),
                 **{k: np.float32(v) if isinstance(v, float) else np.array(v) for k, v in cfg.items()})
        d["u_nom_init"] = opt.u_nom.numpy().copy()
        for t in range(c["steps"]):
            u_prev = np.float32(np.asarray(opt.u).reshape(-1)[0])
            u = ctrl.step(s.copy())
            d[f"s_{t}"] = s.copy(); d[f"u_prev_{t}"] = u_prev
            d[f"noise_{t}"] = rec.raw[-1]
            d[f"u_{t}"] = np.asarray(u, np.float32).reshape(-1)
            d[f"u_nom_{t}"] = opt.u_nom.numpy().copy()
            # J and u_run are returned by predict_and_cost but not stored unless logging: recompute
            # them through the reference's own methods on the recorded tensors
            d[f"opt_ctrl_seq_{t}"] = opt.optimal_control_sequence.copy()
            s = plant_step(plant, s, u)
        # second pass with logging on to capture J / u_run / trajectories (same seed => same draws)
        ctrl2 = make_controller("mppi", cfg, c["pred"])
        ctrl2.controller_logging = True
        ctrl2.optimizer.optimizer_logging = True
        for t in range(c["steps"]):
            u2 = ctrl2.step(d[f"s_{t}"].copy())
            assert np.array_equal(np.asarray(u2, np.float32).reshape(-1), d[f"u_{t}"])
            lv = ctrl2.optimizer.logging_values
            d[f"J_{t}"] = lv["J_logged"].copy()
            d[f"u_run_{t}"] = lv["Q_logged"].copy()
            if c["keep_traj"]:
                d[f"traj_{t}"] = lv["rollout_trajectories_logged"].copy()
        outs = ctrl2.get_outputs()
        assert outs["J_logged"].shape == (c["steps"], c["N"])
        d["steps"] = np.int32(c["steps"])
        np.savez_compressed(os.path.join(out_dir, f"mppi_{name}.npz"), **d)

    # ---- ADAM torch branch (optimizer_rpgd.py:56-82) ------------------------------------------
    adam = ADAM(lib, learning_rate=0.05, beta_1=0.9, beta_2=0.999, epsilon=1e-8)
    adam.build_optimizer(4, 6, (low, high))
    var = torch.tensor(g.uniform(-1, 1, (4, 6, 1)).astype(np.float32))
    ad = dict(var0=var.numpy().copy())
    for t in range(3):
        grad = torch.tensor(g.standard_normal((4, 6, 1)).astype(np.float32))
        var = adam.apply_gradients([(grad, var)])
        ad[f"grad_{t}"] = grad.numpy().copy(); ad[f"var_{t}"] = var.numpy().copy()
    step_count, m_arr, v_arr = adam.get_weights()
    ad.update(step=np.int32(step_count), m=m_arr, v=v_arr)
    np.savez_compressed(os.path.join(out_dir, "adam.npz"), **ad)

    # ---- RPGD (Optimizers/optimizer_rpgd.py via controller_mpc) --------------------------------
    rpgd_cases = {
        "ode_small":  dict(N=16, H=12, p=5, pred="ODE", its=2, steps=4, resamp=2, dist="uniform", seed=5, shift=1),
        "ode_its20":  dict(N=32, H=50, p=10, pred="ODE", its=20, steps=2, resamp=10, dist="uniform", seed=6, shift=1),
        "mlp_cfg4":   dict(N=256, H=50, p=10, pred="MLP", its=20, steps=2, resamp=10, dist="uniform", seed=7, shift=1),
        "ode_normal": dict(N=16, H=10, p=1, pred="ODE", its=3, steps=3, resamp=1, dist="normal", seed=8, shift=2),
    }
    for name, c in rpgd_cases.items():
        cfg = dict(seed=1, mpc_horizon=c["H"], num_rollouts=c["N"], outer_its=c["its"], sample_stdev=0.5,
                   sample_mean=0.0, sample_whole_control_space=True, uniform_dist_min=-1.0, uniform_dist_max=1.0,
                   resamp_per=c["resamp"], period_interpolation_inducing_points=c["p"],
                   SAMPLING_DISTRIBUTION=c["dist"], shift_previous=c["shift"], warmup=False, warmup_iterations=250,
                   learning_rate=0.05, opt_keep_k_ratio=0.25, gradmax_clip=5.0, rtol=1e-3, adam_beta_1=0.9,
                   adam_beta_2=0.999, adam_epsilon=1e-8, mpc_timestep=dt)
        # the rng is created in the ctor and first used by optimizer_reset() inside configure():
        # patch create_rng's product at class level for the duration of construction
        import Control_Toolkit.Optimizers as tmpl
        orig_create = tmpl.create_rng
        holder = {}
        def recording_create(id, seed, computation_library=None):
            holder["rec"] = RecordingRng(orig_create(id, seed, computation_library=computation_library))
            return holder["rec"]
        tmpl.create_rng = recording_create
        try:
            ctrl = make_controller("rpgd", cfg, c["pred"])
        finally:
            tmpl.create_rng = orig_create
        opt, rec = ctrl.optimizer, holder["rec"]
        d = dict(common, low=low, high=high, predictor=np.array(c["pred"]),
                 **{k: (np.float32(v) if isinstance(v, float) else np.array(v)) for k, v in cfg.items()})
        d["reset_draws"] = rec.raw[0]
        d["Q_init"] = opt.Q_tf.detach().numpy().copy()
        plant = O.Predictor(kind="ODE", dt=dt, env=env)
        s = initial_state(c["seed"])
        for t in range(c["steps"]):
            ndraw = len(rec.raw)
            u_prev = np.float32(np.asarray(opt.u).reshape(-1)[0])
            u = ctrl.step(s.copy())
            d[f"s_{t}"] = s.copy(); d[f"u_prev_{t}"] = u_prev
            d[f"u_{t}"] = np.asarray(u, np.float32).reshape(-1)
            if len(rec.raw) > ndraw:
                d[f"resample_draws_{t}"] = rec.raw[-1]
            d[f"Q_{t}"] = opt.Q_tf.detach().numpy().copy()            # warm-started population after the step
            d[f"u_nom_{t}"] = opt.u_nom.detach().numpy().copy()
            stp, m_arr, v_arr = opt.opt.get_weights()
            d[f"adam_step_{t}"] = np.int32(stp); d[f"m_{t}"] = m_arr.copy(); d[f"v_{t}"] = v_arr.copy()
            d[f"ages_{t}"] = opt.trajectory_ages.numpy().copy()
            s = plant_step(plant, s, u)
        d["steps"] = np.int32(c["steps"])
        record_rpgd_logged_outputs(d, lambda: make_controller("rpgd", cfg, c["pred"]), c["steps"])
        np.savez_compressed(os.path.join(out_dir, f"rpgd_{name}.npz"), **d)

    # ---- second environment, C = 2: the SAME unmodified reference optimizers on the planar quadrotor ---------------
    # pins the [N,P,C] / [N,H,C] logic of optimizer_mppi.py / optimizer_rpgd.py / Interpolator.py for C > 1: per-channel
    # interpolation and limits, sums over axes [1,2] (:154-155), clip_by_norm over [1,2] (:315,:334), [C]-shaped u
    import SI_Toolkit.Predictors.predictor_wrapper as pw
    qenv = O.Quad2DParams(terminal_weight=0.4, target_x=0.1)
    inject_quad(qenv, dt)
    pw.ENVIRONMENT = "Quad2D"
    qlow, qhigh = np.array([-1.0, -0.8], np.float32), np.array([1.0, 0.9], np.float32)
    qcommon = dict(environment=np.array("Quad2D"), env_params=qenv.as_array(), env_param_names=np.array(O.QUAD2D_PARAM_NAMES),
                   dt=np.float32(dt), low=qlow, high=qhigh, predictor=np.array("ODE"))

    def quad_state(seed):
        r = np.random.default_rng(seed)
        return np.array([r.uniform(-0.3, 0.3), r.uniform(-0.5, 0.5), r.uniform(0.6, 1.4), r.uniform(-0.5, 0.5),
                         r.uniform(-0.5, 0.5), r.uniform(-1, 1)], np.float32)

    for name, c in {"quad2d": dict(N=128, H=30, p=5, steps=3, seed=21), "quad2d_p1": dict(N=64, H=12, p=1, steps=2, seed=22)}.items():
        cfg = dict(seed=1, mpc_horizon=c["H"], num_rollouts=c["N"], cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0,
                   SQRTRHOINV=0.03, period_interpolation_inducing_points=c["p"], mpc_timestep=dt)
        ctrl = make_controller("mppi", cfg, "ODE", "Quad2D", (qlow, qhigh))
        ctrl.controller_logging = True
        ctrl.optimizer.optimizer_logging = True
        opt = ctrl.optimizer
        rec = RecordingRng(opt.rng); opt.rng = rec
        plant = O.Predictor(kind="ODE", dt=dt, env=qenv)
        s = quad_state(c["seed"])
        d = dict(qcommon, **{k: np.float32(v) if isinstance(v, float) else np.array(v) for k, v in cfg.items()})
        d["u_nom_init"] = opt.u_nom.numpy().copy()
        for t in range(c["steps"]):
            u_prev = np.asarray(opt.u, np.float32).reshape(-1)
            u_prev = np.broadcast_to(u_prev, (2,)).copy()
            u = ctrl.step(s.copy())
            lv = opt.logging_values
            d[f"s_{t}"] = s.copy(); d[f"u_prev_{t}"] = u_prev
            d[f"noise_{t}"] = rec.raw[-1]
            d[f"u_{t}"] = np.asarray(u, np.float32).reshape(-1)
            d[f"u_nom_{t}"] = opt.u_nom.numpy().copy()
            d[f"J_{t}"] = lv["J_logged"].copy(); d[f"u_run_{t}"] = lv["Q_logged"].copy()
            d[f"traj_{t}"] = lv["rollout_trajectories_logged"].copy()
            s = plant_step(plant, s, u)
        d["steps"] = np.int32(c["steps"])
        np.savez_compressed(os.path.join(out_dir, f"mppi_{name}.npz"), **d)

    for name, c in {"quad2d": dict(N=32, H=20, p=5, its=3, steps=4, resamp=2, dist="uniform", seed=23, shift=1),
                    "quad2d_its20": dict(N=64, H=30, p=10, its=20, steps=2, resamp=10, dist="normal", seed=24, shift=1)}.items():
        cfg = dict(seed=1, mpc_horizon=c["H"], num_rollouts=c["N"], outer_its=c["its"], sample_stdev=0.5,
                   sample_mean=0.0, sample_whole_control_space=True, uniform_dist_min=-1.0, uniform_dist_max=1.0,
                   resamp_per=c["resamp"], period_interpolation_inducing_points=c["p"],
                   SAMPLING_DISTRIBUTION=c["dist"], shift_previous=c["shift"], warmup=False, warmup_iterations=250,
                   learning_rate=0.05, opt_keep_k_ratio=0.25, gradmax_clip=5.0, rtol=1e-3, adam_beta_1=0.9,
                   adam_beta_2=0.999, adam_epsilon=1e-8, mpc_timestep=dt)
        import Control_Toolkit.Optimizers as tmpl
        orig_create = tmpl.create_rng
        holder = {}
        def recording_create(id, seed, computation_library=None):
            holder["rec"] = RecordingRng(orig_create(id, seed, computation_library=computation_library))
            return holder["rec"]
        tmpl.create_rng = recording_create
        try:
            ctrl = make_controller("rpgd", cfg, "ODE", "Quad2D", (qlow, qhigh))
        finally:
            tmpl.create_rng = orig_create
        opt, rec = ctrl.optimizer, holder["rec"]
        d = dict(qcommon, **{k: (np.float32(v) if isinstance(v, float) else np.array(v)) for k, v in cfg.items()})
        d["reset_draws"] = rec.raw[0]
        d["Q_init"] = opt.Q_tf.detach().numpy().copy()
        plant = O.Predictor(kind="ODE", dt=dt, env=qenv)
        s = quad_state(c["seed"])
        for t in range(c["steps"]):
            ndraw = len(rec.raw)
            u_prev = np.broadcast_to(np.asarray(opt.u, np.float32).reshape(-1), (2,)).copy()
            u = ctrl.step(s.copy())
            d[f"s_{t}"] = s.copy(); d[f"u_prev_{t}"] = u_prev
            d[f"u_{t}"] = np.asarray(u, np.float32).reshape(-1)
            if len(rec.raw) > ndraw:
                d[f"resample_draws_{t}"] = rec.raw[-1]
            d[f"Q_{t}"] = opt.Q_tf.detach().numpy().copy()
            d[f"u_nom_{t}"] = opt.u_nom.detach().numpy().copy()
            stp, m_arr, v_arr = opt.opt.get_weights()
            d[f"adam_step_{t}"] = np.int32(stp); d[f"m_{t}"] = m_arr.copy(); d[f"v_{t}"] = v_arr.copy()
            d[f"ages_{t}"] = opt.trajectory_ages.numpy().copy()
            s = plant_step(plant, s, u)
        d["steps"] = np.int32(c["steps"])
        record_rpgd_logged_outputs(d, lambda: make_controller("rpgd", cfg, "ODE", "Quad2D", (qlow, qhigh)), c["steps"])
        np.savez_compressed(os.path.join(out_dir, f"rpgd_{name}.npz"), **d)
    # ---- third environment, S + C = 10 (> 8 network inputs), C = 3: the SAME unmodified reference optimizers on the hovercraft ----------
    # ODE and the 10-32-32-7 MLP predictor: pins the C = 3 shapes and, for the MLP, gives the device network a reference-driven fixture
    henv = O.HoverParams(terminal_weight=0.3, target_x=0.2, target_y=-0.1)
    hover_w = O.mlp_default_weights(5, 10, 7)
    inject_hover(henv, dt, hover_w)
    pw.ENVIRONMENT = "Hover"
    hlow, hhigh = np.array([-1.0, -0.7, -1.0], np.float32), np.array([0.9, 0.8, 1.0], np.float32)
    hcommon = dict(environment=np.array("Hover"), env_params=henv.as_array(), env_param_names=np.array(O.HOVER_PARAM_NAMES),
                   dt=np.float32(dt), low=hlow, high=hhigh, mlp_weights=hover_w)

    def hover_state(seed):
        r = np.random.default_rng(seed)
        return np.array([r.uniform(-0.4, 0.4), r.uniform(-0.3, 0.3), r.uniform(-0.4, 0.4), r.uniform(-0.3, 0.3),
                         r.uniform(-0.8, 0.8), r.uniform(-0.5, 0.5), r.uniform(-2, 2)], np.float32)

    for name, c in {"hover_ode": dict(N=96, H=24, p=6, steps=3, seed=31, pred="ODE"), "hover_mlp": dict(N=64, H=16, p=4, steps=3, seed=32, pred="MLP")}.items():
        cfg = dict(seed=1, mpc_horizon=c["H"], num_rollouts=c["N"], cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0,
                   SQRTRHOINV=0.03, period_interpolation_inducing_points=c["p"], mpc_timestep=dt)
        ctrl = make_controller("mppi", cfg, c["pred"], "Hover", (hlow, hhigh))
        ctrl.controller_logging = True
        ctrl.optimizer.optimizer_logging = True
        opt = ctrl.optimizer
        rec = RecordingRng(opt.rng); opt.rng = rec
        plant = O.Predictor(kind="ODE", dt=dt, env=henv)
        s = hover_state(c["seed"])
        d = dict(hcommon, predictor=np.array(c["pred"]), **{k: np.float32(v) if isinstance(v, float) else np.array(v) for k, v in cfg.items()})
        d["u_nom_init"] = opt.u_nom.numpy().copy()
        for t in range(c["steps"]):
            u_prev = np.broadcast_to(np.asarray(opt.u, np.float32).reshape(-1), (3,)).copy()
            u = ctrl.step(s.copy())
            lv = opt.logging_values
            d[f"s_{t}"] = s.copy(); d[f"u_prev_{t}"] = u_prev
            d[f"noise_{t}"] = rec.raw[-1]
            d[f"u_{t}"] = np.asarray(u, np.float32).reshape(-1)
            d[f"u_nom_{t}"] = opt.u_nom.numpy().copy()
            d[f"J_{t}"] = lv["J_logged"].copy(); d[f"u_run_{t}"] = lv["Q_logged"].copy()
            d[f"traj_{t}"] = lv["rollout_trajectories_logged"].copy()
            s = plant_step(plant, s, u)
        d["steps"] = np.int32(c["steps"])
        np.savez_compressed(os.path.join(out_dir, f"mppi_{name}.npz"), **d)

    for name, c in {"hover_ode": dict(N=32, H=16, p=4, its=3, steps=4, resamp=2, dist="uniform", seed=33, shift=1, pred="ODE"),
                    "hover_mlp": dict(N=48, H=14, p=7, its=4, steps=3, resamp=2, dist="normal", seed=34, shift=1, pred="MLP")}.items():
        cfg = dict(seed=1, mpc_horizon=c["H"], num_rollouts=c["N"], outer_its=c["its"], sample_stdev=0.5,
                   sample_mean=0.0, sample_whole_control_space=True, uniform_dist_min=-1.0, uniform_dist_max=1.0,
                   resamp_per=c["resamp"], period_interpolation_inducing_points=c["p"],
                   SAMPLING_DISTRIBUTION=c["dist"], shift_previous=c["shift"], warmup=False, warmup_iterations=250,
                   learning_rate=0.05, opt_keep_k_ratio=0.25, gradmax_clip=5.0, rtol=1e-3, adam_beta_1=0.9,
                   adam_beta_2=0.999, adam_epsilon=1e-8, mpc_timestep=dt)
        import Control_Toolkit.Optimizers as tmpl
        orig_create = tmpl.create_rng
        holder = {}
        def recording_create(id, seed, computation_library=None):
            holder["rec"] = RecordingRng(orig_create(id, seed, computation_library=computation_library))
            return holder["rec"]
        tmpl.create_rng = recording_create
        try:
            ctrl = make_controller("rpgd", cfg, c["pred"], "Hover", (hlow, hhigh))
        finally:
            tmpl.create_rng = orig_create
        opt, rec = ctrl.optimizer, holder["rec"]
        d = dict(hcommon, predictor=np.array(c["pred"]), **{k: (np.float32(v) if isinstance(v, float) else np.array(v)) for k, v in cfg.items()})
        d["reset_draws"] = rec.raw[0]
        d["Q_init"] = opt.Q_tf.detach().numpy().copy()
        plant = O.Predictor(kind="ODE", dt=dt, env=henv)
        s = hover_state(c["seed"])
        for t in range(c["steps"]):
            ndraw = len(rec.raw)
            u_prev = np.broadcast_to(np.asarray(opt.u, np.float32).reshape(-1), (3,)).copy()
            u = ctrl.step(s.copy())
            d[f"s_{t}"] = s.copy(); d[f"u_prev_{t}"] = u_prev
            d[f"u_{t}"] = np.asarray(u, np.float32).reshape(-1)
            if len(rec.raw) > ndraw:
                d[f"resample_draws_{t}"] = rec.raw[-1]
            d[f"Q_{t}"] = opt.Q_tf.detach().numpy().copy()
            d[f"u_nom_{t}"] = opt.u_nom.detach().numpy().copy()
            stp, m_arr, v_arr = opt.opt.get_weights()
            d[f"adam_step_{t}"] = np.int32(stp); d[f"m_{t}"] = m_arr.copy(); d[f"v_{t}"] = v_arr.copy()
            d[f"ages_{t}"] = opt.trajectory_ages.numpy().copy()
            s = plant_step(plant, s, u)
        d["steps"] = np.int32(c["steps"])
        record_rpgd_logged_outputs(d, lambda: make_controller("rpgd", cfg, c["pred"], "Hover", (hlow, hhigh)), c["steps"])
        np.savez_compressed(os.path.join(out_dir, f"rpgd_{name}.npz"), **d)
    pw.ENVIRONMENT = "CartPole"

    # ---- the TF-only optimizers (CEM, random-action, CEM + naive gradient) --------------------------------------------
    envs = {
        "CartPole": dict(env=env, low=low, high=high, C=1, state=initial_state, inject=lambda: inject_constants(env, dt, mlp_w),
                         common=dict(common)),
        "Quad2D": dict(env=qenv, low=qlow, high=qhigh, C=2, state=quad_state, inject=lambda: inject_quad(qenv, dt),
                       common=dict(env_params=qenv.as_array(), env_param_names=np.array(O.QUAD2D_PARAM_NAMES), dt=np.float32(dt))),
        "Hover": dict(env=henv, low=hlow, high=hhigh, C=3, state=hover_state, inject=lambda: inject_hover(henv, dt, hover_w),
                      common=dict(env_params=henv.as_array(), env_param_names=np.array(O.HOVER_PARAM_NAMES), dt=np.float32(dt),
                                  mlp_weights=hover_w)),
    }
    record_tf_only_optimizers(out_dir, cm, envs, dt)

    # ---- a network of another width: the reference names networks by their sizes (config_controllers.yml:8) and hands the name to the
    #      predictor (controller_mpc.py:67-73); the unmodified optimizer_mppi on 5-16-16-4, 5-24-8-4 and 5-64-64-4 tanh MLPs (stand-in predictor: it
    #      takes whatever weight shapes it is given)
    for name, hidden, seed in (("mlp_h16", (16, 16), 71), ("mlp_h24_8", (24, 8), 72), ("mlp_h64", (64, 64), 73)):
        w_small = O.mlp_default_weights(9, 5, 4, hidden)
        inject_constants(env, dt, mlp_w)
        pw.MLP_WEIGHTS = tuple(torch.tensor(a) for a in O.mlp_unpack(w_small, 5, 4, hidden))
        pw.ENVIRONMENT = "CartPole"
        cfg = dict(seed=1, mpc_horizon=25, num_rollouts=96, cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0, SQRTRHOINV=0.03,
                   period_interpolation_inducing_points=5, mpc_timestep=dt)
        spec = f"Dense-5IN-{hidden[0]}H1-{hidden[1]}H2-4OUT-0"
        ctrl = make_controller("mppi", cfg, spec)
        ctrl.controller_logging = True
        ctrl.optimizer.optimizer_logging = True
        opt = ctrl.optimizer
        rec = RecordingRng(opt.rng); opt.rng = rec
        plant = O.Predictor(kind="ODE", dt=dt, env=env)
        s = initial_state(seed)
        d = dict(common, low=low, high=high, predictor=np.array("MLP"), predictor_specification=np.array(spec), hidden_sizes=np.array(hidden, np.int32),
                 **{k: np.float32(v) if isinstance(v, float) else np.array(v) for k, v in cfg.items()})
        d["mlp_weights"] = w_small
        d["u_nom_init"] = opt.u_nom.numpy().copy()
        for t in range(3):
            u_prev = np.float32(np.asarray(opt.u).reshape(-1)[0])
            u = ctrl.step(s.copy())
            lv = opt.logging_values
            d[f"s_{t}"] = s.copy(); d[f"u_prev_{t}"] = u_prev
            d[f"noise_{t}"] = rec.raw[-1]
            d[f"u_{t}"] = np.asarray(u, np.float32).reshape(-1)
            d[f"u_nom_{t}"] = opt.u_nom.numpy().copy()
            d[f"J_{t}"] = lv["J_logged"].copy(); d[f"u_run_{t}"] = lv["Q_logged"].copy()
            d[f"traj_{t}"] = lv["rollout_trajectories_logged"].copy()
            s = plant_step(plant, s, u)
        d["steps"] = np.int32(3)
        np.savez_compressed(os.path.join(out_dir, f"mppi_{name}.npz"), **d)
    inject_constants(env, dt, mlp_w)

    print("golden fixtures written to", out_dir)
    for f in sorted(os.listdir(out_dir)):
        if f.endswith(".npz"):
            print(f"  {f:28s} {os.path.getsize(os.path.join(out_dir, f)) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
