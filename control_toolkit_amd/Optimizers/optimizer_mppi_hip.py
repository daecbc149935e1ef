"""optimizer_mppi_hip — drop-in for reference Optimizers/optimizer_mppi.py (same ctor keys :16-34,
configure :115-139, step :205-225, optimizer_reset :227-231) running on libctk_hip.so."""
from typing import Tuple

import numpy as np

from . import template_optimizer, logging_kwargs
from ..computation_library import HipLibrary


class optimizer_mppi_hip(template_optimizer):
    supported_computation_libraries = (HipLibrary,)
    engine_name = "mppi"

    def __init__(self, predictor, cost_function, control_limits: "Tuple[np.ndarray, np.ndarray]",
                 computation_library, seed, cc_weight: float, R: float, LBD: float, mpc_horizon: int,
                 num_rollouts: int, NU: float, SQRTRHOINV: float, period_interpolation_inducing_points: int,
                 optimizer_logging: bool, calculate_optimal_trajectory: bool = False, **kwargs):
        super().__init__(predictor=predictor, cost_function=cost_function, control_limits=control_limits,
                         optimizer_logging=optimizer_logging, seed=seed, num_rollouts=num_rollouts,
                         mpc_horizon=mpc_horizon, computation_library=computation_library,
                         calculate_optimal_trajectory=calculate_optimal_trajectory,
                         rng_mode=kwargs.get("rng_mode", "device"), device=kwargs.get("device", 0), **logging_kwargs(kwargs))
        self.cc_weight, self.R, self.LBD, self.NU = cc_weight, R, LBD, NU
        self._SQRTRHOINV = SQRTRHOINV
        self.period_interpolation_inducing_points = period_interpolation_inducing_points
        self.global_rollout_offset = int(kwargs.get("global_rollout_offset", 0))
        # optional key of the optimizer entry (swallowed by **kwargs in the reference's constructors): resident_idle_us > 0 serves the steps
        # from a kernel that stays on the device (include/ctk_hip.h: ctk_resident_*); it leaves by itself after that long without a step
        self.resident_idle_us = float(kwargs.get("resident_idle_us", 0.0) or 0.0)

    def configure(self, num_states: int, num_control_inputs: int, dt: float, predictor_specification: str, **kwargs):
        super().configure(num_states=num_states, num_control_inputs=num_control_inputs, default_configure=False)
        self._build_engine(dt, predictor_specification, cc_weight=self.cc_weight, R=self.R, LBD=self.LBD, NU=self.NU,
                           SQRTRHOINV=self._SQRTRHOINV,
                           period_interpolation_inducing_points=self.period_interpolation_inducing_points,
                           global_rollout_offset=self.global_rollout_offset)
        self.number_of_interpolation_inducing_points = self.engine.inducing_points()
        if self.resident_idle_us > 0.0:
            self.engine.resident_enable(True, self.resident_idle_us)
        self.optimizer_reset()

    def step(self, s: np.ndarray, time=None):
        if self.optimizer_logging:
            self.logging_values = {"s_logged": np.asarray(s).copy()}
        s = self._prepare_state(s)
        self._sync_parameters()
        noise = self._draws("normal", [self.num_rollouts, self.number_of_interpolation_inducing_points, self.num_control_inputs])   # :173-175
        self._publish_u(self.engine.step(s, noise, u_prev=self._u_prev()))      # :211-212
        self._lazy.clear()                                                        # u_nom / optimal_control_sequence (:220): read on demand
        if self.optimizer_logging:
            self._fill_logging(s, self.u)
        if self.calculate_optimal_trajectory:                                     # :222-223
            self.optimal_trajectory = self._predict_optimal_trajectory(s, self.u_nom, self._u_prev())
        return self.u

    def optimizer_reset(self):
        self.engine.reset()                                                       # :227-231
        self._lazy.clear()
