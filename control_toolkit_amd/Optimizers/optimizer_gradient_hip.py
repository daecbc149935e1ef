"""optimizer_gradient_hip — drop-in for reference Optimizers/optimizer_gradient_tf.py (ctor keys :16-37,
step :101-173, optimizer_reset :174-185) on libctk_hip.so: Keras-Adam descent on N plans, no resampling."""
from typing import Tuple

import numpy as np

from . import template_optimizer, logging_kwargs
from ..computation_library import HipLibrary


class optimizer_gradient_hip(template_optimizer):
    supported_computation_libraries = (HipLibrary,)
    engine_name = "gradient"

    def __init__(self, predictor, cost_function, control_limits: "Tuple[np.ndarray, np.ndarray]", computation_library,
                 seed, mpc_horizon: int, gradient_steps: int, num_rollouts: int, initial_action_stdev: float,
                 learning_rate: float, adam_beta_1: float, adam_beta_2: float, adam_epsilon: float, gradmax_clip: float,
                 rtol: float, warmup: bool, warmup_iterations: int, optimizer_logging: bool,
                 calculate_optimal_trajectory: bool = False, **kwargs):
        super().__init__(predictor=predictor, cost_function=cost_function, control_limits=control_limits,
                         optimizer_logging=optimizer_logging, seed=seed, num_rollouts=num_rollouts,
                         mpc_horizon=mpc_horizon, computation_library=computation_library,
                         calculate_optimal_trajectory=calculate_optimal_trajectory,
                         rng_mode=kwargs.get("rng_mode", "device"), device=kwargs.get("device", 0), **logging_kwargs(kwargs))
        self.gradient_steps = gradient_steps
        self.initial_action_stdev = initial_action_stdev     # declared by the reference, unused there too (:52)
        self.learning_rate = learning_rate
        self.adam_beta_1, self.adam_beta_2, self.adam_epsilon = adam_beta_1, adam_beta_2, adam_epsilon
        self.gradmax_clip, self.rtol = gradmax_clip, rtol
        self.warmup, self.warmup_iterations = warmup, warmup_iterations
        self.count = 0

    def configure(self, num_states: int, num_control_inputs: int, dt: float = None, predictor_specification=None, **kwargs):
        super().configure(num_states=num_states, num_control_inputs=num_control_inputs, default_configure=False)
        if dt is None:
            raise ValueError("optimizer_gradient_hip.configure needs dt")
        self._build_engine(dt, predictor_specification, outer_its=self.gradient_steps, learning_rate=self.learning_rate,
                           adam_beta_1=self.adam_beta_1, adam_beta_2=self.adam_beta_2, adam_epsilon=self.adam_epsilon,
                           gradmax_clip=self.gradmax_clip, warmup=int(bool(self.warmup)),
                           warmup_iterations=self.warmup_iterations)
        self.optimizer_reset()

    def step(self, s: np.ndarray, time=None):
        if self.optimizer_logging:
            self.logging_values = {"s_logged": np.asarray(s).copy()}
        s = self._prepare_state(s)
        self._sync_parameters()
        tail = self._draws("uniform", [self.num_rollouts, 1, self.num_control_inputs])          # :137-142
        u_prev = self._u_prev()
        self.u = np.squeeze(self.engine.step(s, tail, u_prev=u_prev))
        if self.optimizer_logging:                                         # :135-140
            self.logging_values["Q_logged"] = self._logged("Q")
            self.logging_values["J_logged"] = self._logged("J")
            self.logging_values["u_logged"] = self.u
        self.count += 1
        return self.u

    def optimizer_reset(self):
        self.engine.reset(self._draws("uniform", [self.num_rollouts, self.mpc_horizon, self.num_control_inputs]))   # :174-185
        self.count = 0
