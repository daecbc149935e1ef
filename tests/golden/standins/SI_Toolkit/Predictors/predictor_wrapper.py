"""Stand-in PredictorWrapper: the build's own plants in torch fp32 (differentiable, so the reference's RPGD can
backpropagate through them) — cart-pole ODE / 5-32-32-4 tanh MLP, the planar quadrotor ODE (6 states, 2 inputs) and the hovercraft with a
reaction wheel (7 states, 3 inputs: ODE and its 10-32-32-7 tanh MLP).
Same formulas as oracle/ctk_oracle.py:Predictor; constants are injected by tests/golden/make_golden.py."""
import torch

# injected by make_golden.py ------------------------------------------------------------------
ENVIRONMENT = "CartPole"   # which plant the next PredictorWrapper() models
CONSTANTS = {}      # derived fp32 constants (oracle.derived_constants)
MLP_WEIGHTS = None  # tuple of torch tensors (W1,b1,W2,b2,W3,b3)


class PredictorWrapper:
    def __init__(self):
        self.environment = ENVIRONMENT
        self.num_states, self.num_control_inputs = {"Quad2D": (6, 2), "Hover": (7, 3)}.get(ENVIRONMENT, (4, 1))
        self.kind = None
        self.batch_size = None

    def configure(self, batch_size, dt=None, computation_library=None, variable_parameters=None,
                  predictor_specification=None, horizon=None, **kwargs):
        self.batch_size = batch_size
        self.kind = predictor_specification
        self.dt = dt

    def copy(self):
        return PredictorWrapper()

    def update(self, s=None, Q0=None):
        pass   # no recurrent state

    def _quad_step(self, s, q):
        k = CONSTANTS
        x, vx, z, vz, th, om = s.unbind(1)
        aF = k["g"] + k["kF"] * (q[:, 0] + q[:, 1])
        aM = k["kM"] * (q[:, 0] - q[:, 1])
        sn, cs = torch.sin(th), torch.cos(th)
        ax = -aF * sn - k["c_v"] * vx
        az = aF * cs - k["g"] - k["c_v"] * vz
        al = aM - k["c_w"] * om
        dt = k["dt"]
        return torch.stack([x + dt * vx, vx + dt * ax, z + dt * vz, vz + dt * az, th + dt * om, om + dt * al], 1)

    def _hover_step(self, s, q):
        k = CONSTANTS
        x, vx, y, vy, th, om, w = s.unbind(1)
        fb, fl = k["aF"] * q[:, 0], k["aL"] * q[:, 1]
        sn, cs = torch.sin(th), torch.cos(th)
        ax = fb * cs - fl * sn - k["c_v"] * vx
        ay = fb * sn + fl * cs - k["c_v"] * vy
        al = -k["kT"] * q[:, 2] - k["c_w"] * om
        aw = k["kW"] * q[:, 2] - k["c_ww"] * w
        dt = k["dt"]
        return torch.stack([x + dt * vx, vx + dt * ax, y + dt * vy, vy + dt * ay, th + dt * om, om + dt * al, w + dt * aw], 1)

    def _mlp(self, s, q):
        W1, b1, W2, b2, W3, b3 = MLP_WEIGHTS
        xin = torch.cat([s, q], 1)
        return torch.tanh(torch.tanh(xin @ W1.T + b1) @ W2.T + b2) @ W3.T + b3

    def _step(self, s, q):
        if self.environment == "Quad2D":
            return self._quad_step(s, q)
        if self.environment == "Hover":
            return self._hover_step(s, q) if self.kind == "ODE" else self._mlp(s, q)
        q = q[:, 0]
        if self.kind == "ODE":
            k = CONSTANTS
            x, v, th, om = s.unbind(1)
            sn, cs = torch.sin(th), torch.cos(th)
            A = k["u_max"] * q + k["k_ml"] * om * om * sn - k["M_fric"] * v
            tmp = A * k["inv_mt"]
            D = k["k43l"] - k["k_mpl_mt"] * cs * cs
            Nn = k["g"] * sn - cs * tmp - k["k_jf"] * om
            thdd = Nn / D
            xdd = tmp - k["k_mpl_mt"] * thdd * cs
            return torch.stack([x + k["dt"] * v, v + k["dt"] * xdd, th + k["dt"] * om, om + k["dt"] * thdd], 1)
        W1, b1, W2, b2, W3, b3 = MLP_WEIGHTS
        xin = torch.cat([s, q[:, None]], 1)
        h1 = torch.tanh(xin @ W1.T + b1)
        h2 = torch.tanh(h1 @ W2.T + b2)
        return h2 @ W3.T + b3

    def predict_core(self, s, Q):
        states = [s]
        cur = s
        for h in range(Q.shape[1]):
            cur = self._step(cur, Q[:, h, :])
            states.append(cur)
        return torch.stack(states, 1)
