"""optimizer_rpgd_hip — drop-in for reference Optimizers/optimizer_rpgd.py (ctor keys :148-179,
configure :247-273, step :388-524, optimizer_reset :527-548) running on libctk_hip.so."""
from typing import Tuple

import numpy as np

from . import template_optimizer, logging_kwargs
from ..computation_library import HipLibrary


class optimizer_rpgd_hip(template_optimizer):
    supported_computation_libraries = (HipLibrary,)
    engine_name = "rpgd"

    def __init__(self, predictor, cost_function, control_limits: "Tuple[np.ndarray, np.ndarray]", computation_library,
                 seed, mpc_horizon: int, num_rollouts: int, outer_its: int, sample_stdev: float, sample_mean: float,
                 sample_whole_control_space: bool, uniform_dist_min: float, uniform_dist_max: float, resamp_per: int,
                 period_interpolation_inducing_points: int, SAMPLING_DISTRIBUTION: str, shift_previous: int,
                 warmup: bool, warmup_iterations: int, learning_rate: float, opt_keep_k_ratio: float,
                 gradmax_clip: float, rtol: float, adam_beta_1: float, adam_beta_2: float, adam_epsilon: float,
                 optimizer_logging: bool, calculate_optimal_trajectory: bool = False, **kwargs):
        super().__init__(predictor=predictor, cost_function=cost_function, control_limits=control_limits,
                         optimizer_logging=optimizer_logging, seed=seed, num_rollouts=num_rollouts,
                         mpc_horizon=mpc_horizon, computation_library=computation_library,
                         calculate_optimal_trajectory=calculate_optimal_trajectory,
                         rng_mode=kwargs.get("rng_mode", "device"), device=kwargs.get("device", 0), **logging_kwargs(kwargs))
        self.outer_its = outer_its
        self.sample_stdev, self.sample_mean = sample_stdev, sample_mean
        self.sample_whole_control_space = sample_whole_control_space
        # reference :200-206: the whole control space means [action_low[c], action_high[c]] per input
        self.sample_min, self.sample_max = uniform_dist_min, uniform_dist_max
        self.resamp_per = resamp_per
        self.period_interpolation_inducing_points = period_interpolation_inducing_points
        self.shift_previous = shift_previous
        self.do_warmup, self.warmup_iterations = warmup, warmup_iterations
        self.opt_keep_k = int(max(int(num_rollouts * opt_keep_k_ratio), 1))   # :213
        self.gradmax_clip, self.rtol = gradmax_clip, rtol
        if SAMPLING_DISTRIBUTION not in ("normal", "uniform"):
            raise ValueError(f"RPGD cannot interpret sampling type {SAMPLING_DISTRIBUTION}")   # :291
        self.SAMPLING_DISTRIBUTION = SAMPLING_DISTRIBUTION
        self.first_iter_count = warmup_iterations if warmup else outer_its    # :219-221
        self.learning_rate = learning_rate
        self.adam_beta_1, self.adam_beta_2, self.adam_epsilon = adam_beta_1, adam_beta_2, adam_epsilon
        # The reference picks the update rule with the computation library (optimizer_rpgd.py:35-53): TensorFlow wraps
        # tf.keras.optimizers.Adam (the YAML entry `rpgd-tf`), PyTorch runs the in-repo Adam (:56-82).  Here it is a key of its
        # own: adam_rule: torch (default) | keras.
        self.adam_rule = kwargs.get("adam_rule", "torch")
        if self.adam_rule not in ("torch", "keras"):
            raise ValueError(f"adam_rule must be 'torch' or 'keras', got {self.adam_rule!r}")
        self.summed_stage_cost = None
        self.count = 0

    def configure(self, num_states: int, num_control_inputs: int, **kwargs):
        dt = kwargs.get("dt", None)
        predictor_specification = kwargs.get("predictor_specification", None)
        super().configure(num_states=num_states, num_control_inputs=num_control_inputs, default_configure=False)
        if dt is None or predictor_specification is None:
            raise ValueError("RPGD requires dt and predictor_specification to be passed.")   # :271
        self._build_engine(
            dt, predictor_specification, outer_its=self.outer_its, resamp_per=self.resamp_per,
            shift_previous=self.shift_previous, opt_keep_k=self.opt_keep_k,
            sampling_distribution=0 if self.SAMPLING_DISTRIBUTION == "uniform" else 1,
            sample_whole_control_space=int(bool(self.sample_whole_control_space)),
            sample_stdev=self.sample_stdev, sample_mean=self.sample_mean, sample_min=self.sample_min,
            sample_max=self.sample_max, learning_rate=self.learning_rate, gradmax_clip=self.gradmax_clip,
            adam_beta_1=self.adam_beta_1, adam_beta_2=self.adam_beta_2, adam_epsilon=self.adam_epsilon,
            adam_rule=1 if self.adam_rule == "keras" else 0,
            warmup=int(bool(self.do_warmup)), warmup_iterations=self.warmup_iterations,
            period_interpolation_inducing_points=self.period_interpolation_inducing_points)
        self.number_of_interpolation_inducing_points = self.engine.inducing_points()
        self.optimizer_reset()

    def _sample_kind(self):
        return "normal" if self.SAMPLING_DISTRIBUTION == "normal" else "uniform"

    def step(self, s: np.ndarray, time=None):
        if self.optimizer_logging:
            self.logging_values = {"s_logged": np.asarray(s).copy()}
        s = self._prepare_state(s)
        self._sync_parameters()
        need = self.engine.samples_needed()
        draws = None
        if need:
            draws = self._draws(self._sample_kind(), [self.num_rollouts - self.opt_keep_k,
                                                      self.number_of_interpolation_inducing_points, self.num_control_inputs])
        u_prev = self._u_prev()
        u = self.engine.step(s, draws, u_prev=u_prev)
        self._lazy.clear()                                           # u_nom (:426), optimal_control_sequence (:435): read on demand
        if self.optimizer_logging:                                   # :428-433
            # get_action's rollout of the descended plans (:340-343,:424): the engine materialises it when logging (ctk_api.hip: rpgd_materialize)
            self.rollout_trajectories = self._logged("TRAJ")
            self.logging_values["Q_logged"] = self._logged("Q")
            self.logging_values["J_logged"] = self._logged("J")
            self.logging_values["rollout_trajectories_logged"] = self.rollout_trajectories
            self.logging_values["trajectory_ages_logged"] = self._logged("AGES")
            self.logging_values["u_logged"] = self.u
        self.count += 1
        if self.calculate_optimal_trajectory:                        # :518-521
            self.optimal_trajectory, self.summed_stage_cost = self._predict_optimal_trajectory(s, self.u_nom, u_prev, want_summed_stage_cost=True)
        self.u = np.asarray(u, np.float32).reshape(-1).copy()        # :523
        return self.u

    @property
    def trajectory_ages(self):
        return self.engine.read("AGES")

    def optimizer_reset(self):
        draws = self._draws(self._sample_kind(), [self.num_rollouts, self.number_of_interpolation_inducing_points, self.num_control_inputs])
        self.engine.reset(draws)                                     # :527-548
        self._lazy.clear()
        self.count = 0
