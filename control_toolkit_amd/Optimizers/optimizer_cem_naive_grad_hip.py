"""optimizer_cem_naive_grad_hip — drop-in for reference Optimizers/optimizer_cem_naive_grad_tf.py (ctor keys
:14-31, step :89-115, optimizer_reset :117-119) on libctk_hip.so."""
from typing import Tuple

import numpy as np

from . import template_optimizer, logging_kwargs
from ..computation_library import HipLibrary


class optimizer_cem_naive_grad_hip(template_optimizer):
    supported_computation_libraries = (HipLibrary,)
    engine_name = "cem_naive_grad"

    def __init__(self, predictor, cost_function, control_limits: "Tuple[np.ndarray, np.ndarray]", computation_library,
                 seed, mpc_horizon: int, cem_outer_it: int, num_rollouts: int, cem_initial_action_stdev: float,
                 cem_stdev_min: float, cem_best_k: int, learning_rate: float, gradmax_clip: float,
                 optimizer_logging: bool, calculate_optimal_trajectory: bool = False, **kwargs):
        super().__init__(predictor=predictor, cost_function=cost_function, control_limits=control_limits,
                         optimizer_logging=optimizer_logging, seed=seed, num_rollouts=num_rollouts,
                         mpc_horizon=mpc_horizon, computation_library=computation_library,
                         calculate_optimal_trajectory=calculate_optimal_trajectory,
                         rng_mode=kwargs.get("rng_mode", "device"), device=kwargs.get("device", 0), **logging_kwargs(kwargs))
        self.cem_outer_it = cem_outer_it
        self.cem_initial_action_stdev, self.cem_stdev_min, self.cem_best_k = cem_initial_action_stdev, cem_stdev_min, cem_best_k
        self.learning_rate, self.gradmax_clip = learning_rate, gradmax_clip

    def configure(self, num_states: int, num_control_inputs: int, dt: float = None, predictor_specification=None, **kwargs):
        super().configure(num_states=num_states, num_control_inputs=num_control_inputs, default_configure=False)
        if dt is None:
            raise ValueError("optimizer_cem_naive_grad_hip.configure needs dt")
        self._build_engine(dt, predictor_specification, cem_outer_it=self.cem_outer_it, cem_best_k=self.cem_best_k,
                           cem_initial_action_stdev=self.cem_initial_action_stdev, cem_stdev_min=self.cem_stdev_min,
                           learning_rate=self.learning_rate, gradmax_clip=self.gradmax_clip)
        self.optimizer_reset()

    def step(self, s: np.ndarray, time=None):
        if self.optimizer_logging:
            self.logging_values = {"s_logged": np.asarray(s).copy()}
        s = self._prepare_state(s)
        self._sync_parameters()
        noise = self._draws("normal", [self.cem_outer_it, self.num_rollouts, self.mpc_horizon, self.num_control_inputs])
        u_prev = self._u_prev()
        self.u = np.squeeze(self.engine.step(s, noise, u_prev=u_prev))
        if self.optimizer_logging:                                         # :106-110
            self.logging_values["Q_logged"] = self._logged("Q")
            self.logging_values["J_logged"] = self._logged("J")
            self.logging_values["u_logged"] = self.u
        return self.u

    @property
    def dist_mue(self):
        return self.engine.read("U_NOM")

    @property
    def stdev(self):
        return self.engine.read("STD")

    def optimizer_reset(self):
        self.engine.reset()
