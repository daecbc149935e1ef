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
        if optimizer_name in {None, ""}:
            optimizer_name = str(self.config_controller["optimizer"])
        if predictor_specification in {None, ""}:
            predictor_specification = self.config_controller.get("predictor_specification", None)
        if self._config_optimizers is None:
            self._config_optimizers = load_yaml(os.path.join("Control_Toolkit_ASF", "config_optimizers.yml"))
        config_optimizer = self._config_optimizers[optimizer_name]
        cost_function_specification = self.config_controller.get("cost_function_specification", None)
        self.cost_function = self._cost_function or CostFunctionWrapper()        # :40
        self.predictor = self._predictor or PredictorWrapper()                   # :43
        Optimizer = import_optimizer_by_name(optimizer_name)                     # :56
        opt_kwargs = dict(config_optimizer)
        if self.device is not None and "device" not in opt_kwargs:
            opt_kwargs["device"] = self.lib.device_ordinal(self.device)
        self.optimizer = Optimizer(                                              # :57-65
            predictor=self.predictor, cost_function=self.cost_function, control_limits=self.control_limits,
            optimizer_logging=self.controller_logging, computation_library=self.computation_library,
            calculate_optimal_trajectory=self.config_controller.get("calculate_optimal_trajectory"),
            **opt_kwargs)
        self.predictor.configure(batch_size=self.optimizer.num_rollouts, dt=config_optimizer["mpc_timestep"],   # :67-73
                                 computation_library=self.computation_library,
                                 variable_parameters=self.variable_parameters,
                                 predictor_specification=predictor_specification)
        self.cost_function.configure(batch_size=self.optimizer.num_rollouts, horizon=self.optimizer.mpc_horizon,  # :75-82
                                     variable_parameters=self.variable_parameters,
                                     environment_name=self.environment_name,
                                     computation_library=self.computation_library,
                                     cost_function_specification=cost_function_specification)
        self.optimizer.configure(dt=config_optimizer["mpc_timestep"], predictor_specification=predictor_specification,  # :84-89
                                 num_states=self.predictor.num_states,
                                 num_control_inputs=self.predictor.num_control_inputs)
        self.controller_data_for_csv = self.cost_function.cost_function.logged_attributes   # :91
        self.step = self.lib.set_device(self.config_controller.get("device", "gpu"))(self.step)   # :93-96

    def step(self, s: np.ndarray, time=None, updated_attributes: dict = {}):
        self.cost_function.update_cost_parameters_from_config()   # :101
        self.update_attributes(updated_attributes)                # :103
        u = self.optimizer.step(s, time)                          # :104
        self.update_logs(self.optimizer.logging_values)           # :105
        return u

    def controller_reset(self):
        self.optimizer.optimizer_reset()                          # :108-109
