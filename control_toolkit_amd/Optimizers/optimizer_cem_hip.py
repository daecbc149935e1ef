"""optimizer_cem_hip — drop-in for reference Optimizers/optimizer_cem_tf.py (ctor keys :16-34,
step :83-111, optimizer_reset :113-117) running on libctk_hip.so."""
from typing import Tuple

import numpy as np

from . import template_optimizer, logging_kwargs
from ..computation_library import HipLibrary


class optimizer_cem_hip(template_optimizer):
    supported_computation_libraries = (HipLibrary,)
    engine_name = "cem"

    def __init__(self, predictor, cost_function, control_limits: "Tuple[np.ndarray, np.ndarray]",
                 computation_library, seed, mpc_horizon: int, cem_outer_it: int, cem_initial_action_stdev: float,
                 num_rollouts: int, cem_stdev_min: float, cem_best_k: int, warmup: bool, warmup_iterations: int,
                 optimizer_logging: bool, calculate_optimal_trajectory: bool = False, **kwargs):
        super().__init__(predictor=predictor, cost_function=cost_function, control_limits=control_limits,
                         optimizer_logging=optimizer_logging, seed=seed, num_rollouts=num_rollouts,
                         mpc_horizon=mpc_horizon, computation_library=computation_library,
                         calculate_optimal_trajectory=calculate_optimal_trajectory,
                         rng_mode=kwargs.get("rng_mode", "device"), device=kwargs.get("device", 0), **logging_kwargs(kwargs))
        self.cem_outer_it = cem_outer_it
        self.cem_initial_action_stdev = cem_initial_action_stdev
        self.cem_stdev_min = cem_stdev_min
        self.cem_best_k = cem_best_k
        self.warmup = warmup
        self.warmup_iterations = warmup_iterations
        self.count = 0

    def configure(self, num_states: int, num_control_inputs: int, dt: float = None, predictor_specification=None, **kwargs):
        super().configure(num_states=num_states, num_control_inputs=num_control_inputs, default_configure=False)
        if dt is None:
            raise ValueError("optimizer_cem_hip.configure needs dt")
        self._build_engine(dt, predictor_specification, cem_outer_it=self.cem_outer_it, cem_best_k=self.cem_best_k,
                           warmup=int(bool(self.warmup)), warmup_iterations=self.warmup_iterations,
                           cem_initial_action_stdev=self.cem_initial_action_stdev, cem_stdev_min=self.cem_stdev_min)
        self.optimizer_reset()

    def step(self, s: np.ndarray, time=None):
        if self.optimizer_logging:
            self.logging_values = {"s_logged": np.asarray(s).copy()}
        s = self._prepare_state(s)
        self._sync_parameters()
        iterations = self.warmup_iterations if self.warmup and self.count == 0 else self.cem_outer_it   # :92
        noise = self._draws("normal", [iterations, self.num_rollouts, self.mpc_horizon, self.num_control_inputs])   # :64-65
        self._publish_u(self.engine.step(s, noise, u_prev=self._u_prev()))
        if self.optimizer_logging:
            self._fill_logging(s, self.u)
        self.count += 1
        return self.u

    @property
    def dist_mue(self):
        return self.engine.read("U_NOM")

    @property
    def stdev(self):
        return self.engine.read("STD")

    def optimizer_reset(self):
        self.engine.reset()
        self.count = 0
        self.u = 0.0
