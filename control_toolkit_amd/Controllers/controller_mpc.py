"""controller_mpc — mirror of reference Controllers/controller_mpc.py:21-109: wires predictor,
optimizer and cost in the same order and exposes the same step()."""
import os
from typing import Optional

import numpy as np

from . import template_controller, load_yaml
from ..Cost_Functions import CostFunctionWrapper
from ..Predictors import PredictorWrapper
from ..others.globals_and_utils import get_logger, import_optimizer_by_name

logger = get_logger(__name__)


class controller_mpc(template_controller):
    _has_optimizer = True

    def __init__(self, environment_name, control_limits, initial_environment_attributes,
                 config_controllers: dict = None, config_optimizers: dict = None, predictor: PredictorWrapper = None,
                 cost_function: CostFunctionWrapper = None):
        super().__init__(environment_name, control_limits, initial_environment_attributes, config_controllers)
        # reference :16 loads Control_Toolkit_ASF/config_optimizers.yml at import time
        self._config_optimizers = config_optimizers
        self._predictor = predictor
        self._cost_function = cost_function

    def configure(self, optimizer_name: Optional[str] = None, predictor_specification: Optional[str] = None):
        """Same wiring order as the reference (:24-96): shells of cost function and predictor first, then the
        optimizer built around them, then each of the three configured with what only the others know."""
        cc = self.config_controller
        optimizer_name = str(cc["optimizer"]) if optimizer_name in {None, ""} else optimizer_name
        if predictor_specification in {None, ""}:
            predictor_specification = cc.get("predictor_specification", None)
        if self._config_optimizers is None:                                       # reference :16 reads this file at import time
            self._config_optimizers = load_yaml(os.path.join("Control_Toolkit_ASF", "config_optimizers.yml"))
        opt_cfg = self._config_optimizers[optimizer_name]
        dt = opt_cfg["mpc_timestep"]                                              # :69,:85
        self.cost_function = self._cost_function if self._cost_function is not None else CostFunctionWrapper(environment_name=self.environment_name)   # :40
        self.predictor = self._predictor if self._predictor is not None else PredictorWrapper(environment_name=self.environment_name)                   # :43
        self.optimizer = self._make_optimizer(optimizer_name, opt_cfg)            # :56-65
        N, H = self.optimizer.num_rollouts, self.optimizer.mpc_horizon
        shared = dict(computation_library=self.computation_library, variable_parameters=self.variable_parameters)
        self.predictor.configure(batch_size=N, dt=dt, predictor_specification=predictor_specification, **shared)   # :67-73
        self.cost_function.configure(batch_size=N, horizon=H, environment_name=self.environment_name,               # :75-82
                                     cost_function_specification=cc.get("cost_function_specification", None), **shared)
        self.optimizer.configure(dt=dt, predictor_specification=predictor_specification,                            # :84-89
                                 num_states=self.predictor.num_states, num_control_inputs=self.predictor.num_control_inputs)
        self.controller_data_for_csv = self.cost_function.cost_function.logged_attributes                           # :91
        self.step = self.lib.set_device(cc.get("device", "gpu"))(self.step)                                         # :93-96

    def _make_optimizer(self, optimizer_name: str, opt_cfg: dict):
        kwargs = dict(opt_cfg)
        if self.device is not None:
            kwargs.setdefault("device", self.lib.device_ordinal(self.device))
        cls = import_optimizer_by_name(optimizer_name)                            # discovery by file name (:56)
        return cls(predictor=self.predictor, cost_function=self.cost_function, control_limits=self.control_limits,
                   optimizer_logging=self.controller_logging, computation_library=self.computation_library,
                   calculate_optimal_trajectory=self.config_controller.get("calculate_optimal_trajectory"), **kwargs)

    def step(self, s: np.ndarray, time=None, updated_attributes: dict = {}):
        self.cost_function.update_cost_parameters_from_config()   # :101
        self.update_attributes(updated_attributes)                # :103
        u = self.optimizer.step(s, time)                          # :104
        self.update_logs(self.optimizer.logging_values)           # :105
        return u

    def controller_reset(self):
        self.optimizer.optimizer_reset()                          # :108-109
