#!/usr/bin/env python3
"""Build-container probe (run by tests/test_reference_plugin_cpu.py in a subprocess): the UNMODIFIED reference
`controller_mpc` (read from /root/reference) wires predictor, cost function and optimizer itself
(Controllers/controller_mpc.py:24-96), resolves `optimizer: mppi-hip` through its own discovery
(others/globals_and_utils.py:103-133: glob by file name under Control_Toolkit_ASF/Optimizers/) and drives the plug-in
through `step` (:99-106).  The engine is replaced by a recording stub — no GPU here — so what is checked is the
plug-in boundary: constructor / configure / step signatures, the reference-shaped predictor and cost objects, the
values that reach the engine.  The reference's third-party imports are satisfied by tests/golden/standins (see there).
Prints one JSON line."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as MG   # noqa: E402  (setup_workdir: temp dir with ./Control_Toolkit -> /root/reference and the stand-ins)

work = MG.setup_workdir()
os.makedirs(os.path.join("Control_Toolkit_ASF", "Optimizers"), exist_ok=True)
open(os.path.join("Control_Toolkit_ASF", "Optimizers", "__init__.py"), "w").close()
for name in ("mppi", "rpgd"):     # the one-line shims of INTEGRATION.md section 3
    with open(os.path.join("Control_Toolkit_ASF", "Optimizers", f"optimizer_{name}_hip.py"), "w") as f:
        f.write(f"from control_toolkit_amd.Optimizers.optimizer_{name}_hip import optimizer_{name}_hip  # noqa: F401\n")
import yaml   # noqa: E402
yaml.safe_dump({"mpc": {"optimizer": "mppi-hip", "predictor_specification": "ODE", "cost_function_specification": "default",
                        "computation_library": "numpy", "controller_logging": False, "calculate_optimal_trajectory": False,
                        "device": "cpu"}}, open(os.path.join("Control_Toolkit_ASF", "config_controllers.yml"), "w"))
yaml.safe_dump({"mppi-hip": dict(seed=7, mpc_horizon=25, num_rollouts=96, cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0, SQRTRHOINV=0.03,
                                 period_interpolation_inducing_points=5, mpc_timestep=0.02,
                                 predictor_parameters={"m_pole": 0.1}, predictor_intermediate_steps=2),
                "rpgd-hip": dict(seed=7, mpc_horizon=20, num_rollouts=16, outer_its=2, sample_stdev=0.5, sample_mean=0.0,
                                 sample_whole_control_space=True, uniform_dist_min=-1.0, uniform_dist_max=1.0, resamp_per=10,
                                 period_interpolation_inducing_points=5, SAMPLING_DISTRIBUTION="uniform", shift_previous=1, warmup=False,
                                 warmup_iterations=0, learning_rate=0.05, opt_keep_k_ratio=0.25, gradmax_clip=5.0, rtol=1e-3,
                                 adam_beta_1=0.9, adam_beta_2=0.999, adam_epsilon=1e-8, mpc_timestep=0.02)},
               open(os.path.join("Control_Toolkit_ASF", "config_optimizers.yml"), "w"))
yaml.safe_dump({"cost_function_name_default": "default", "CartPole": {"default": {"dd_weight": 123.0, "ep_weight": 4567.0}}},
               open(os.path.join("Control_Toolkit_ASF", "config_cost_function.yml"), "w"))

import control_toolkit_amd.Optimizers as T   # noqa: E402
from control_toolkit_amd._capi import environment_info   # noqa: E402

CALLS = []


class StubEngine:
    """records what the plug-in asks of the engine (control_toolkit_amd/_capi.py:CtkEngine surface)"""

    def __init__(self, optimizer, predictor, **kw):
        self.S, self.C, self.param_names = environment_info(kw.get("environment", "CartPole"))
        self.environment = kw.get("environment", "CartPole")
        self.N, self.H = kw["num_rollouts"], kw["mpc_horizon"]
        self.P = int(np.ceil((self.H - 1) / kw.get("period_interpolation_inducing_points", 1)) + 1)
        self.params = {}
        CALLS.append(("create", optimizer, predictor, {k: (np.asarray(v).tolist() if isinstance(v, np.ndarray) else v) for k, v in kw.items()}))

    def set_param(self, name, value): self.params[name] = float(value)
    def get_param(self, name): return self.params[name]
    def inducing_points(self): return self.P
    def samples_needed(self): return 0
    def samples_needed_reset(self): return self.N * self.P * self.C
    def reset(self, draws=None): CALLS.append(("reset", None if draws is None else list(np.shape(draws))))
    def read(self, name): return np.zeros((1, self.H, self.C), np.float32)

    def step(self, s, samples=None, u_prev=None, loc=None):
        CALLS.append(("step", np.asarray(s).tolist(), None if samples is None else list(np.shape(samples)), np.asarray(u_prev).tolist()))
        return np.full(self.C, 0.25, np.float32)


T.CtkEngine = StubEngine

import Control_Toolkit.Controllers.controller_mpc as cm   # noqa: E402  — the reference module, unmodified
out = {}
low, high = np.array([-1.0], np.float32), np.array([1.0], np.float32)
ctrl = cm.controller_mpc("CartPole", (low, high), {"target_position": 0.05})
ctrl.configure()
opt = ctrl.optimizer
out["optimizer_class"] = type(opt).__name__
out["optimizer_module"] = type(opt).__module__
out["lib"] = type(ctrl.computation_library).__name__
out["predictor_class"] = type(ctrl.predictor).__module__ + "." + type(ctrl.predictor).__name__
out["cost_class"] = type(ctrl.cost_function).__module__ + "." + type(ctrl.cost_function).__name__
u = ctrl.step(np.array([0.0, 0.1, 0.2, 0.3], np.float32))
out["u"] = np.asarray(u).tolist()
ctrl.step(np.array([0.0, 0.1, 0.2, 0.3], np.float32), updated_attributes={"target_position": -0.3})
out["params_after_update"] = dict(opt.engine.params)
ctrl.controller_reset()
out["calls"] = CALLS[:]
# the RPGD plug-in through the same reference wiring
CALLS.clear()
ctrl2 = cm.controller_mpc("CartPole", (low, high), {})
ctrl2.configure(optimizer_name="rpgd-hip")
ctrl2.step(np.zeros(4, np.float32))
out["rpgd_class"] = type(ctrl2.optimizer).__name__
out["rpgd_calls"] = CALLS[:]
print("PROBE_JSON " + json.dumps(out))
