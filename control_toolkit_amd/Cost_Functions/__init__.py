"""Cost side of the boundary.  `cost_function_base` semantics (reference
Cost_Functions/__init__.py:38-93: MAX_COST shift, zero default terminal cost, mean over H+1)
are implemented inside the rollout kernels; this wrapper carries the cost PARAMETERS and the
hot-reload flag of reference Cost_Functions/cost_function_wrapper.py:71-74."""

DEFAULT_COST = dict(dd_weight=600.0, ep_weight=20000.0, ekp_weight=80.0, cc_weight=1.0, ccrc_weight=1.0, R=1.0,
                    x_scale=0.198, terminal_weight=0.0)
DEFAULT_ATTRIBUTES = dict(target_position=0.0, target_equilibrium=1.0)


class CostFunctionWrapper:
    MAX_COST = 0.0

    def __init__(self, parameters=None):
        self.parameters = dict(DEFAULT_COST)
        if parameters:
            unknown = set(parameters) - set(DEFAULT_COST)
            if unknown:
                raise ValueError(f"unknown cost parameters {sorted(unknown)}")
            self.parameters.update(parameters)
        self.reload_cost_parameters_from_config_flag = False
        self.logged_attributes = {}
        self.version = 0          # bumped whenever parameters change; optimizers re-upload
        self.batch_size = self.horizon = None
        self.variable_parameters = None
        self.cost_function = self  # reference: wrapper.cost_function.logged_attributes (controller_mpc.py:91)

    def configure(self, batch_size, horizon, variable_parameters=None, environment_name=None,
                  computation_library=None, cost_function_specification=None):
        self.batch_size, self.horizon = batch_size, horizon
        self.variable_parameters = variable_parameters
        self.environment_name = environment_name
        self.cost_function_specification = cost_function_specification

    def set_parameters(self, **kw):
        unknown = set(kw) - set(DEFAULT_COST)
        if unknown:
            raise ValueError(f"unknown cost parameters {sorted(unknown)}")
        self.parameters.update(kw)
        self.reload_cost_parameters_from_config_flag = True

    def update_cost_parameters_from_config(self):
        # reference cost_function_wrapper.py:71-74 (flag set by the YAML watchdog thread)
        if self.reload_cost_parameters_from_config_flag:
            self.version += 1
            self.reload_cost_parameters_from_config_flag = False

    def copy(self):
        return CostFunctionWrapper(self.parameters)
