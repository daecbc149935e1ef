"""Cost side of the boundary.  `cost_function_base` semantics (reference
Cost_Functions/__init__.py:38-93: MAX_COST shift, zero default terminal cost, mean over H+1)
are implemented inside the rollout kernels; this wrapper carries the cost PARAMETERS — the declarative
description the kernels take as constants — and the hot reload of reference
Cost_Functions/cost_function_wrapper.py:56-74 + CostFunctionUpdater.py end to end:

    config_cost_function.yml  --(watcher thread sets a flag)-->  controller_mpc.step():
        update_cost_parameters_from_config()  -->  optimizer._sync_parameters()  -->  ctk_set_param

The YAML has the reference's layout (Control_Toolkit_ASF_Template/config_cost_function.yml):

    cost_function_name_default: default
    CartPole:
      default:
        dd_weight: 600.0      # any subset of DEFAULT_COST; unknown keys are an error (the kernels implement
        ep_weight: 20000.0    # exactly these terms: oracle/ctk_oracle.py:Cost)
"""
import os

from .CostFunctionUpdater import CostFunctionUpdater

# cost parameters of the built environments (the cost section of the library's parameter lists, ctk_param_name): the
# concrete cost classes the reference would import from Control_Toolkit_ASF.Cost_Functions.<environment>.<name>
# (cost_function_wrapper.py:59-66) are kernels here, so what a "cost function" carries is this dictionary
DEFAULT_COST_BY_ENV = {
    "CartPole": dict(dd_weight=600.0, ep_weight=20000.0, ekp_weight=80.0, cc_weight=1.0, ccrc_weight=1.0, R=1.0,
                     x_scale=0.198, terminal_weight=0.0),
    "Quad2D": dict(pos_weight=400.0, ang_weight=150.0, vel_weight=8.0, angvel_weight=1.5, cc_weight=1.0, ccrc_weight=2.0, R=1.0,
                   pos_scale=0.5, terminal_weight=0.0),
    "Hover": dict(pos_weight=300.0, ang_weight=80.0, vel_weight=6.0, angvel_weight=1.0, wheel_weight=0.02, cc_weight=1.0, ccrc_weight=1.5, R=1.0,
                  pos_scale=0.5, terminal_weight=0.0),
}
DEFAULT_ATTRIBUTES_BY_ENV = {"CartPole": dict(target_position=0.0, target_equilibrium=1.0), "Quad2D": dict(target_x=0.0, target_z=1.0),
                             "Hover": dict(target_x=0.0, target_y=0.0)}
DEFAULT_COST = DEFAULT_COST_BY_ENV["CartPole"]
DEFAULT_ATTRIBUTES = DEFAULT_ATTRIBUTES_BY_ENV["CartPole"]
DEFAULT_CONFIG_PATH = os.path.join("Control_Toolkit_ASF", "config_cost_function.yml")   # reference cost_function_wrapper.py:14


def _checked(section, where, allowed=None):
    allowed = DEFAULT_COST if allowed is None else allowed
    if not isinstance(section, dict):
        raise ValueError(f"{where}: expected a mapping of cost parameters")
    unknown = set(section) - set(allowed)
    if unknown:
        raise ValueError(f"{where}: unknown cost parameters {sorted(unknown)} (built: {sorted(allowed)})")
    return {k: float(v) for k, v in section.items()}


class CostFunctionWrapper:
    MAX_COST = 0.0

    def __init__(self, parameters=None, config_path=None, watch: bool = True, environment_name: str = "CartPole"):
        """parameters: explicit values (highest precedence at construction).  config_path: the cost YAML; if
        None and Control_Toolkit_ASF/config_cost_function.yml exists in the working directory (the reference's
        CWD-relative convention) that file is used.  watch: start the CostFunctionUpdater thread in configure().
        environment_name: which built environment's cost this is (configure() takes it from the controller,
        reference controller_mpc.py:75-82)."""
        from ..Predictors import built_environment
        self.environment_name = built_environment(environment_name)
        self._allowed = DEFAULT_COST_BY_ENV[self.environment_name]
        self.parameters = dict(self._allowed)
        if parameters:
            self.parameters.update(_checked(parameters, "CostFunctionWrapper(parameters=...)", self._allowed))
        self._explicit = dict(parameters or {})
        self.config_path = config_path if config_path is not None else (DEFAULT_CONFIG_PATH if os.path.isfile(DEFAULT_CONFIG_PATH) else None)
        self.watch = watch
        self.config = None            # the YAML section in force (set by configure / the updater)
        self.cost_function_name_default = "default"
        self.cost_function_name = None
        self.cost_function_updater = None
        self.reload_cost_parameters_from_config_flag = False
        self.logged_attributes = {}
        self.version = 0          # bumped whenever parameters change
        self.batch_size = self.horizon = None
        self.variable_parameters = None
        self.cost_function = self  # reference: wrapper.cost_function.logged_attributes (controller_mpc.py:91)

    # reference cost_function_wrapper.py:76-88
    def update_cost_function_name_from_specification(self, cost_function_specification=None):
        if cost_function_specification is None:
            self.cost_function_name = self.cost_function_name_default.replace("-", "_")
        elif isinstance(cost_function_specification, str):
            self.cost_function_name = cost_function_specification.replace("-", "_")
        else:
            raise ValueError(f"Cannot interpret cost function specification {cost_function_specification}.")

    def configure(self, batch_size, horizon, variable_parameters=None, environment_name=None,
                  computation_library=None, cost_function_specification=None):
        self.batch_size, self.horizon = batch_size, horizon
        self.variable_parameters = variable_parameters
        if environment_name is not None:
            from ..Predictors import built_environment
            env = built_environment(environment_name)
            if env != self.environment_name:           # the controller decides; explicit values must fit the new set
                self.environment_name, self._allowed = env, DEFAULT_COST_BY_ENV[env]
                self.parameters = dict(self._allowed, **_checked(self._explicit, "CostFunctionWrapper(parameters=...)", self._allowed))
        self.yaml_environment_name = environment_name if environment_name is not None else self.environment_name
        self.cost_function_specification = cost_function_specification
        if self.config_path is not None:
            from yaml import safe_load
            whole = safe_load(open(self.config_path, "r")) or {}
            self.cost_function_name_default = str(whole.get("cost_function_name_default", "default"))
            self.update_cost_function_name_from_specification(cost_function_specification)
            environment_name = self.yaml_environment_name
            try:
                section = whole[environment_name][self.cost_function_name]
            except (KeyError, TypeError):
                raise KeyError(f"{self.config_path}: no section [{environment_name}][{self.cost_function_name}]") from None
            self.config = section
            self.parameters.update(_checked(section, f"{self.config_path}[{environment_name}][{self.cost_function_name}]", self._allowed))
            self.parameters.update(self._explicit)
            self.cost_function_updater = CostFunctionUpdater(self, environment_name, self.cost_function_name,   # reference :69
                                                             start_thread=self.watch)
        else:
            self.update_cost_function_name_from_specification(cost_function_specification)

    def set_parameters(self, **kw):
        """Programmatic equivalent of editing the YAML: takes effect at the next controller step."""
        self.config = dict(self.config or {}, **_checked(kw, "set_parameters", self._allowed))
        for k in kw:                        # an explicit later choice replaces the constructor's for that parameter
            self._explicit.pop(k, None)
        self.reload_cost_parameters_from_config_flag = True

    def validate_config(self, section):
        """raises ValueError for a section this cost function cannot take (CostFunctionUpdater.poll_now)"""
        _checked(section, "cost YAML reload", self._allowed)

    def reload_cost_parameters_from_config(self):
        # the reference's cost functions re-read their attributes from self.config here; constructor-supplied
        # parameters keep their precedence over the file, as at configure()
        self.parameters.update(_checked(self.config or {}, "cost YAML reload", self._allowed))
        self.parameters.update(self._explicit)
        self.version += 1

    def update_cost_parameters_from_config(self):
        # reference cost_function_wrapper.py:71-74 (flag set by the watcher thread)
        if self.reload_cost_parameters_from_config_flag:
            self.reload_cost_parameters_from_config_flag = False
            self.reload_cost_parameters_from_config()

    def copy(self):
        c = CostFunctionWrapper(self.parameters, watch=False, environment_name=self.environment_name)
        c.config_path = None          # a copy carries values, it does not watch (one watcher per path)
        c.cost_function_name = self.cost_function_name
        return c
