"""template_controller — mirror of reference Controllers/__init__.py:27-178 (config lookup,
computation-library switch, variable parameters, log stacking)."""
import os
from abc import ABC, abstractmethod
from types import SimpleNamespace
from typing import Tuple

import numpy as np
import yaml

from ..computation_library import ComputationClasses, HipLibrary


def load_yaml(path):
    with open(path, "r") as f:
        return yaml.safe_load(f)


class VariableParameters(SimpleNamespace):
    """stand for SI_Toolkit.General.variable_parameters.VariableParameters (external)."""

    def __init__(self, lib=None):
        super().__init__()
        self.lib = lib

    def set_attributes(self, attributes, device=None):
        for k, v in attributes.items():
            setattr(self, k, v)

    def update_attributes(self, attributes):
        for k, v in attributes.items():
            setattr(self, k, v)


class template_controller(ABC):
    _has_optimizer = False
    _computation_library = None

    def __init__(self, environment_name: str, control_limits: "Tuple[np.ndarray, np.ndarray]",
                 initial_environment_attributes: dict, config_controllers: dict = None):
        # reference :39-43: Control_Toolkit_ASF/config_controllers.yml[<controller name>] (CWD-relative);
        # a dict may be passed instead so that no file layout is required
        if config_controllers is None:
            config_controllers = load_yaml(os.path.join("Control_Toolkit_ASF", "config_controllers.yml"))
        self.config_controller = dict(config_controllers[self.controller_name])
        name = str(self.config_controller.get("computation_library", ""))
        if name:
            if "hip" in name.lower():
                self._computation_library = HipLibrary()
            else:   # reference :46-58 raises for unknown names; TF/torch/numpy are not built here
                raise ValueError(f"Computation library {name} could not be interpreted (this build provides 'hip').")
        elif not isinstance(self.computation_library, ComputationClasses):
            raise ValueError(f"{self.__class__.__name__} does not have a default computation library set.")
        self.environment_name = environment_name
        self.control_limits = control_limits
        self.action_low, self.action_high = self.control_limits
        device = str(self.config_controller["device"]) if "device" in self.config_controller else None
        if device is not None:
            self.configure = self.lib.set_device(device)(self.configure)
        self.device = device
        self.initial_environment_attributes = {k: self.lib.to_variable(v, self.lib.float32)
                                               for k, v in initial_environment_attributes.items()}
        self.variable_parameters = VariableParameters(self.lib)
        self.variable_parameters.set_attributes(self.initial_environment_attributes, device=device)
        self.u = 0.0
        self.controller_logging = self.config_controller["controller_logging"]
        self.save_vars = ["Q_logged", "J_logged", "s_logged", "u_logged", "realized_cost_logged",
                          "trajectory_ages_logged", "rollout_trajectories_logged"]
        self.logs = {s: [] for s in self.save_vars}
        self.controller_data_for_csv = {}

    def configure(self, **kwargs):
        pass

    def update_attributes(self, updated_attributes: dict):
        self.variable_parameters.update_attributes(updated_attributes)

    @abstractmethod
    def step(self, s: np.ndarray, time=None, updated_attributes: dict = {}):
        return None

    def controller_reset(self):
        raise NotImplementedError

    @property
    def controller_name(self):
        name = self.__class__.__name__
        if name != "template_controller":
            return name.replace("controller_", "").replace("_", "-").lower()
        raise AttributeError()

    @property
    def computation_library(self):
        if self._computation_library is None:
            raise NotImplementedError("Controller class needs to specify its computation library")
        return self._computation_library

    @property
    def lib(self):
        return self.computation_library

    @property
    def has_optimizer(self):
        return self._has_optimizer

    @staticmethod
    def _stack(v):
        from ..Optimizers import DeviceLogEntry
        if len(v) == 0:
            return None
        if all(isinstance(x, DeviceLogEntry) for x in v):
            return DeviceLogEntry.gather(v)          # one transfer per run of consecutive steps
        arrs = [x.numpy() if isinstance(x, DeviceLogEntry) else np.asarray(x) for x in v]
        if len({a.shape for a in arrs}) > 1:     # RPGD logs self.u BEFORE updating it (optimizer_rpgd.py:433,523): 0.0, then [u]
            arrs = [a.reshape(-1) for a in arrs]
        return np.stack(arrs, axis=0)

    def get_outputs(self):
        # reference :159-168
        return {name: self._stack(v) for name, v in self.logs.items()}

    def update_logs(self, logging_values: dict) -> None:
        # reference :170-178 (logged arrays are copies).  Entries that live in the engine's HBM log ring
        # (optimizer option logging_on_device) stay there: the handle is kept and the data moves in get_outputs();
        # when the ring is about to wrap, what it holds is brought to the host first.
        if self.controller_logging:
            from ..Optimizers import DeviceLogEntry
            for name in self.save_vars:
                var = logging_values.get(name, None)
                if var is None:
                    continue
                if isinstance(var, DeviceLogEntry):
                    log = self.logs[name]
                    pending = [x for x in log if isinstance(x, DeviceLogEntry)]
                    if pending and var.step - pending[0].step + 1 >= var.engine.log_capacity:
                        host = DeviceLogEntry.gather(pending)
                        it = iter(host)
                        self.logs[name] = log = [next(it) if isinstance(x, DeviceLogEntry) else x for x in log]
                    log.append(var)
                else:
                    self.logs[name].append(np.array(var, copy=True))
