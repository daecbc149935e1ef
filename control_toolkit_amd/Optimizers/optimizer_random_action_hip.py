"""optimizer_random_action_hip — drop-in for reference Optimizers/optimizer_random_action_tf.py
(ctor keys :15-27, step :49-76) running on libctk_hip.so."""
from typing import Tuple

import numpy as np

from . import template_optimizer, logging_kwargs
from ..computation_library import HipLibrary


class optimizer_random_action_hip(template_optimizer):
    supported_computation_libraries = (HipLibrary,)
    engine_name = "random_action"

    def __init__(self, predictor, cost_function, control_limits: "Tuple[np.ndarray, np.ndarray]",
                 computation_library, seed, mpc_horizon: int, num_rollouts: int, optimizer_logging: bool,
                 calculate_optimal_trajectory: bool = False, **kwargs):
        super().__init__(predictor=predictor, cost_function=cost_function, control_limits=control_limits,
                         optimizer_logging=optimizer_logging, seed=seed, num_rollouts=num_rollouts,
                         mpc_horizon=mpc_horizon, computation_library=computation_library,
                         calculate_optimal_trajectory=calculate_optimal_trajectory,
                         rng_mode=kwargs.get("rng_mode", "device"), device=kwargs.get("device", 0), **logging_kwargs(kwargs))

    def configure(self, num_states: int, num_control_inputs: int, dt: float = None, predictor_specification=None, **kwargs):
        super().configure(num_states=num_states, num_control_inputs=num_control_inputs, default_configure=False)
        if dt is None:
            raise ValueError("optimizer_random_action_hip.configure needs dt")
        self._build_engine(dt, predictor_specification)
        self.optimizer_reset()

    def step(self, s: np.ndarray, time=None):
        if self.optimizer_logging:
            self.logging_values = {"s_logged": np.asarray(s).copy()}
        s = self._prepare_state(s)
        self._sync_parameters()
        u01 = self._draws("uniform", [self.num_rollouts, self.mpc_horizon, self.num_control_inputs])   # :56-61
        self._publish_u(self.engine.step(s, u01, u_prev=self._u_prev()))
        if self.optimizer_logging:
            self._fill_logging(s, self.u)
        return self.u

    def optimizer_reset(self):
        self.engine.reset()
