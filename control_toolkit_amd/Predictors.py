"""PredictorWrapper — host-side stand for SI_Toolkit.Predictors.predictor_wrapper.PredictorWrapper
(external to the reference; call sites optimizer_mppi.py:188, controller_mpc.py:43,67-73).  On
MI355X the rollout is fused into the optimizer kernels, so this object only carries the
predictor *specification* (kind, dt, physical parameters, network weights) to the engine."""
import numpy as np

DEFAULT_DYNAMICS = dict(g=9.81, m_cart=0.230, m_pole=0.087, L=0.1975, u_max=2.62, M_fric=4.77, J_fric=2.5e-4)
MLP_NUM_WEIGHTS = 1380   # W1[32,5] b1[32] W2[32,32] b2[32] W3[4,32] b3[4]
GRU_NUM_WEIGHTS = 10212  # per layer W_i[96,I] W_h[96,32] b_i[96] b_h[96] (I = 5, 32), then W_o[4,32] b_o[4]


class PredictorWrapper:
    def __init__(self, parameters=None, weights=None, intermediate_steps: int = 1):
        self.num_states = 4
        self.num_control_inputs = 1
        self.parameters = dict(DEFAULT_DYNAMICS)
        if parameters:
            unknown = set(parameters) - set(DEFAULT_DYNAMICS)
            if unknown:
                raise ValueError(f"unknown dynamics parameters {sorted(unknown)}")
            self.parameters.update(parameters)
        self.weights = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32).ravel()
        self.intermediate_steps = int(intermediate_steps)
        self.predictor_specification = None
        self.batch_size = None
        self.dt = None

    def configure(self, batch_size, dt=None, computation_library=None, variable_parameters=None,
                  predictor_specification=None, horizon=None, **kwargs):
        spec = "ODE" if predictor_specification in (None, "") else str(predictor_specification)
        up = spec.upper()
        kind = ("ODE" if up.startswith("ODE") else "GRU" if up.startswith("GRU")      # 'GRU-6IN-32H1-32H2-5OUT-0' convention
                else "MLP" if up.startswith("MLP") or up.startswith("DENSE") else None)
        if kind is None:
            raise NotImplementedError(f"predictor_specification {spec!r}: only 'ODE', 'MLP' and 'GRU' are built")
        if kind in ("MLP", "GRU"):
            want = MLP_NUM_WEIGHTS if kind == "MLP" else GRU_NUM_WEIGHTS
            if self.weights is None:
                raise ValueError(f"{kind} predictor needs weights (PredictorWrapper(weights=...))")
            if self.weights.size != want:
                raise ValueError(f"{kind} predictor expects {want} weights, got {self.weights.size}")
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
            weights_file: net.npy | net.npz (key `weights`)              # flat fp32 network weights (MLP 1380 / GRU 10212)
        `section`: optional top-level key (e.g. the environment name).  Weight files are read with
        numpy.load(allow_pickle=False) only."""
        import os
        from yaml import safe_load
        cfg = safe_load(open(path, "r")) or {}
        if section is not None:
            cfg = cfg[section]
        unknown = set(cfg) - {"dynamics", "intermediate_steps", "weights_file"}
        if unknown:
            raise ValueError(f"{path}: unknown predictor keys {sorted(unknown)}")
        weights = None
        if cfg.get("weights_file"):
            wf = cfg["weights_file"]
            wf = wf if os.path.isabs(wf) else os.path.join(os.path.dirname(os.path.abspath(path)), wf)
            data = np.load(wf, allow_pickle=False)
            weights = data["weights"] if hasattr(data, "files") else data
        return cls(cfg.get("dynamics"), weights, int(cfg.get("intermediate_steps", 1)))

    def copy(self):
        return PredictorWrapper(self.parameters, self.weights, self.intermediate_steps)

    def update(self, s=None, Q0=None):
        """RNN hidden-state advance in the reference (optimizer_mppi.py:195-197).  The carried state lives
        on the device behind the optimizer's engine (ctk_predictor_update); the MPPI step advances it itself,
        so this host object has nothing to do."""
        return None
