"""optimizer_cem_grad_bharadhwaj_hip — drop-in for reference Optimizers/optimizer_cem_grad_bharadhwaj_tf.py (ctor keys
:17-38, step :151-178, optimizer_reset :180-184) on libctk_hip.so."""
from typing import Tuple

import numpy as np

from . import template_optimizer, logging_kwargs
from ..computation_library import HipLibrary


class optimizer_cem_grad_bharadhwaj_hip(template_optimizer):
    supported_computation_libraries = (HipLibrary,)
    engine_name = "cem_grad_bharadhwaj"

    def __init__(self, predictor, cost_function, control_limits: "Tuple[np.ndarray, np.ndarray]", computation_library,
                 seed, mpc_horizon: int, num_rollouts: int, cem_outer_it: int, cem_initial_action_stdev: float,
                 cem_stdev_min: float, cem_best_k: int, learning_rate: float, adam_beta_1: float, adam_beta_2: float,
                 adam_epsilon: float, gradmax_clip: float, warmup: bool, warmup_iterations: int, optimizer_logging: bool,
                 calculate_optimal_trajectory: bool = False, **kwargs):
        super().__init__(predictor=predictor, cost_function=cost_function, control_limits=control_limits,
                         optimizer_logging=optimizer_logging, seed=seed, num_rollouts=num_rollouts,
                         mpc_horizon=mpc_horizon, computation_library=computation_library,
                         calculate_optimal_trajectory=calculate_optimal_trajectory,
                         rng_mode=kwargs.get("rng_mode", "device"), device=kwargs.get("device", 0), **logging_kwargs(kwargs))
        self.cem_outer_it, self.cem_best_k = cem_outer_it, cem_best_k
        self.cem_initial_action_stdev, self.cem_stdev_min = cem_initial_action_stdev, cem_stdev_min
        self.learning_rate, self.gradmax_clip = learning_rate, gradmax_clip
        self.adam_beta_1, self.adam_beta_2, self.adam_epsilon = adam_beta_1, adam_beta_2, adam_epsilon
        self.warmup, self.warmup_iterations = warmup, warmup_iterations
        self.count = 0

    def configure(self, num_states: int, num_control_inputs: int, dt: float = None, predictor_specification=None, **kwargs):
        super().configure(num_states=num_states, num_control_inputs=num_control_inputs, default_configure=False)
        if dt is None:
            raise ValueError("optimizer_cem_grad_bharadhwaj_hip.configure needs dt")
        self._build_engine(dt, predictor_specification, cem_outer_it=self.cem_outer_it, cem_best_k=self.cem_best_k,
                           cem_initial_action_stdev=self.cem_initial_action_stdev, cem_stdev_min=self.cem_stdev_min,
                           learning_rate=self.learning_rate, gradmax_clip=self.gradmax_clip,
                           adam_beta_1=self.adam_beta_1, adam_beta_2=self.adam_beta_2, adam_epsilon=self.adam_epsilon,
                           warmup=int(bool(self.warmup)), warmup_iterations=self.warmup_iterations)
        self.optimizer_reset()

    def step(self, s: np.ndarray, time=None):
        if self.optimizer_logging:
            self.logging_values = {"s_logged": np.asarray(s).copy()}
        s = self._prepare_state(s)
        self._sync_parameters()
        iterations = self.warmup_iterations if self.warmup and self.count == 0 else self.cem_outer_it   # :161
        draws = None
        if not getattr(self.rng, "on_device", False):
            H, K, N = self.mpc_horizon, self.cem_best_k, self.num_rollouts
            el = self._draws("normal", [K, H, self.num_control_inputs])                                    # :158
            rest = self._draws("normal", [iterations, N - K, H, self.num_control_inputs])                  # :94
            draws = np.concatenate([el.ravel(), rest.ravel()]).astype(np.float32)
        u_prev = self._u_prev()
        self.u = np.squeeze(self.engine.step(s, draws, u_prev=u_prev))
        if self.optimizer_logging:                                                   # :170-175
            self.logging_values["Q_logged"] = self._logged("Q")
            self.logging_values["J_logged"] = self._logged("J")
            self.logging_values["u_logged"] = self.u
        self.count += 1
        return self.u

    @property
    def dist_mue(self):
        return self.engine.read("U_NOM")

    @property
    def stdev(self):
        return self.engine.read("STD")

    def optimizer_reset(self):
        self.engine.reset()                                                          # :180-184 (the Adam state is NOT reset)
        self.count = 0
