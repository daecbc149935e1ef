"""PredictorWrapper — host-side stand for SI_Toolkit.Predictors.predictor_wrapper.PredictorWrapper
(external to the reference; call sites optimizer_mppi.py:188, controller_mpc.py:43,67-73).  On
MI355X the rollout is fused into the optimizer kernels, so this object only carries the
predictor *specification* (kind, dt, physical parameters, network weights) to the engine."""
import numpy as np

# The environments built into libctk_hip.so (include/ctk_hip.h: ctk_environment): dimensions and the dynamics section of
# their parameter lists (the cost section lives in Cost_Functions).  Static here so that describing a predictor does not
# load the library; tests/test_host_cpu.py checks the table against ctk_env_info / ctk_param_name.
ENVIRONMENT_DIMS = {"CartPole": (4, 1), "Quad2D": (6, 2), "Hover": (7, 3)}
DEFAULT_DYNAMICS_BY_ENV = {
    "CartPole": dict(g=9.81, m_cart=0.230, m_pole=0.087, L=0.1975, u_max=2.62, M_fric=4.77, J_fric=2.5e-4),
    "Quad2D": dict(g=9.81, mass=0.5, inertia=0.004, arm=0.12, thrust_gain=0.6, drag_lin=0.25, drag_ang=0.4),
    "Hover": dict(mass=1.2, inertia=0.05, wheel_inertia=0.01, thrust_max=4.0, lateral_max=1.5, torque_max=0.2, drag_lin=0.3, drag_ang=0.2,
                  wheel_friction=0.05),
}
DEFAULT_DYNAMICS = DEFAULT_DYNAMICS_BY_ENV["CartPole"]
MLP_NUM_WEIGHTS = 1380   # CartPole: W1[32,5] b1[32] W2[32,32] b2[32] W3[4,32] b3[4]
GRU_NUM_WEIGHTS = 10212  # CartPole: per layer W_i[96,I] W_h[96,32] b_i[96] b_h[96] (I = 5, 32), then W_o[4,32] b_o[4]


MAX_HIDDEN = {"MLP": 64, "GRU": 32}   # widest hidden layer built per network type (include/ctk_hip.h: cfg.predictor_hidden1/2); narrower layers are embedded exactly


def network_weight_count(kind: str, num_states: int, num_control_inputs: int, hidden=(32, 32)) -> int:
    """flat fp32 weights of a (S+C)-h1-h2-S tanh MLP, or of two GRU layers (h1, h2 units) + dense h2->S"""
    I, S = num_states + num_control_inputs, num_states
    h1, h2 = int(hidden[0]), int(hidden[1])
    if kind == "MLP":
        return I * h1 + h1 + h1 * h2 + h2 + h2 * S + S
    if kind == "GRU":
        return (3 * h1 * I + 3 * h1 * h1 + 6 * h1) + (3 * h2 * h1 + 3 * h2 * h2 + 6 * h2) + (h2 * S + S)
    return 0


def parse_predictor_specification(predictor_specification):
    """(kind, sizes) from what `controller_mpc.configure` hands the predictor (reference Controllers/controller_mpc.py:67-73).
    kind: 'ODE' | 'MLP' | 'GRU'.  sizes: None, or dict(inputs, h1, h2, outputs) when the string follows the reference's network-name
    convention `<Type>-<I>IN-<h1>H1-<h2>H2-<O>OUT-<n>` (Control_Toolkit_ASF_Template/config_controllers.yml:8:
    `GRU-6IN-32H1-32H2-5OUT-0`; `Dense-...` for a feed-forward net)."""
    import re
    spec = "ODE" if predictor_specification in (None, "") else str(predictor_specification)
    up = spec.upper()
    kind = ("ODE" if up.startswith("ODE") else "GRU" if up.startswith("GRU") else "MLP" if up.startswith("MLP") or up.startswith("DENSE") else None)
    if kind is None:
        raise NotImplementedError(f"predictor_specification {spec!r}: only 'ODE', 'MLP' / 'Dense' and 'GRU' networks are built")
    sizes = None
    m = re.search(r"-(\d+)IN((?:-\d+H\d+)+)-(\d+)OUT", up)
    if m and kind != "ODE":
        hidden = [int(x) for x in re.findall(r"-(\d+)H\d+", m.group(2))]
        if len(hidden) != 2:
            raise NotImplementedError(f"network {spec!r}: {len(hidden)} hidden layers; the predictor kernels are built for two")
        sizes = dict(inputs=int(m.group(1)), h1=hidden[0], h2=hidden[1], outputs=int(m.group(3)))
    return kind, sizes


def check_network_sizes(spec, sizes, num_states: int, num_control_inputs: int, kind: str = None):
    """the sizes a network name states against the environment and the kernels; returns (h1, h2)"""
    if sizes is None:
        return (32, 32)
    if kind is None:
        kind = parse_predictor_specification(spec)[0]
    widest = MAX_HIDDEN.get(kind, 32)
    I, S = num_states + num_control_inputs, num_states
    if sizes["inputs"] != I or sizes["outputs"] != S:
        raise ValueError(f"network {spec!r} has {sizes['inputs']} inputs / {sizes['outputs']} outputs; this environment's predictor maps "
                         f"{I} (states + control inputs) to {S} (next state)")
    if max(sizes["h1"], sizes["h2"]) > widest or min(sizes["h1"], sizes["h2"]) < 1:
        raise NotImplementedError(f"network {spec!r}: hidden widths {sizes['h1']} / {sizes['h2']}; the {kind} predictor kernels hold up to {widest} units "
                                  f"per hidden layer (narrower layers are embedded exactly, wider ones are not built)")
    return (sizes["h1"], sizes["h2"])


def built_environment(name) -> str:
    stem = str(name or "CartPole").replace("-", "").replace("_", "").lower()
    for built in ENVIRONMENT_DIMS:                  # (user environments are added here by build_env.register_environment)
        if stem.startswith(built.replace("_", "").lower()):
            return built
    raise NotImplementedError(f"environment {name!r} is not built (have: {sorted(ENVIRONMENT_DIMS)})")


class PredictorWrapper:
    def __init__(self, parameters=None, weights=None, intermediate_steps: int = 1, environment_name: str = "CartPole"):
        self.environment_name = built_environment(environment_name)
        self.num_states, self.num_control_inputs = ENVIRONMENT_DIMS[self.environment_name]
        self.parameters = dict(DEFAULT_DYNAMICS_BY_ENV[self.environment_name])
        if parameters:
            unknown = set(parameters) - set(self.parameters)
            if unknown:
                raise ValueError(f"unknown dynamics parameters {sorted(unknown)} for environment {self.environment_name}")
            self.parameters.update(parameters)
        self.weights = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32).ravel()
        self.intermediate_steps = int(intermediate_steps)
        self.predictor_specification = None
        self.batch_size = None
        self.dt = None

    def configure(self, batch_size, dt=None, computation_library=None, variable_parameters=None,
                  predictor_specification=None, horizon=None, **kwargs):
        spec = "ODE" if predictor_specification in (None, "") else str(predictor_specification)
        kind, sizes = parse_predictor_specification(spec)       # 'GRU-6IN-32H1-32H2-5OUT-0' convention: the name carries the sizes
        self.hidden_sizes = (32, 32)
        if kind in ("MLP", "GRU"):
            self.hidden_sizes = check_network_sizes(spec, sizes, self.num_states, self.num_control_inputs, kind)
            want = network_weight_count(kind, self.num_states, self.num_control_inputs, self.hidden_sizes)
            if self.weights is None:
                raise ValueError(f"{kind} predictor needs weights (PredictorWrapper(weights=...))")
            if self.weights.size != want:
                raise ValueError(f"{kind} predictor {spec!r} expects {want} weights ({self.num_states + self.num_control_inputs}-"
                                 f"{self.hidden_sizes[0]}-{self.hidden_sizes[1]}-{self.num_states}), got {self.weights.size}")
        self.kind = kind
        self.predictor_specification = spec
        self.batch_size = batch_size
        self.dt = dt
        self.variable_parameters = variable_parameters

    @classmethod
    def from_yaml(cls, path, section=None):
        """Declarative predictor description -> kernel constants.  YAML keys (all optional):
            dynamics: {g, m_cart, m_pole, L, u_max, M_fric, J_fric}     # the ODE's physical parameters
            intermediate_steps: 1                                        # Euler sub-steps per mpc_timestep
            weights_file: net.npy | net.npz (key `weights`)              # flat fp32 network weights (CartPole: MLP 1380 / GRU 10212)
            environment_name: CartPole | Quad2D                          # default: `section` if it names a built environment, else CartPole
        `section`: optional top-level key (e.g. the environment name).  Weight files are read with
        numpy.load(allow_pickle=False) only."""
        import os
        from yaml import safe_load
        cfg = safe_load(open(path, "r")) or {}
        if section is not None:
            cfg = cfg[section]
        unknown = set(cfg) - {"dynamics", "intermediate_steps", "weights_file", "environment_name"}
        if unknown:
            raise ValueError(f"{path}: unknown predictor keys {sorted(unknown)}")
        weights = None
        if cfg.get("weights_file"):
            wf = cfg["weights_file"]
            wf = wf if os.path.isabs(wf) else os.path.join(os.path.dirname(os.path.abspath(path)), wf)
            data = np.load(wf, allow_pickle=False)
            weights = data["weights"] if hasattr(data, "files") else data
        env = cfg.get("environment_name")
        if env is None:
            try:
                env = built_environment(section)
            except NotImplementedError:
                env = "CartPole"
        return cls(cfg.get("dynamics"), weights, int(cfg.get("intermediate_steps", 1)), env)

    def copy(self):
        return PredictorWrapper(self.parameters, self.weights, self.intermediate_steps, self.environment_name)

    def update(self, s=None, Q0=None):
        """RNN hidden-state advance in the reference (optimizer_mppi.py:195-197).  The carried state lives
        on the device behind the optimizer's engine (ctk_predictor_update); the MPPI step advances it itself,
        so this host object has nothing to do."""
        return None
