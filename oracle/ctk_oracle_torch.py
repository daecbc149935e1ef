"""TEST INFRASTRUCTURE — multi-threaded CPU restatement of the MPPI step (torch-CPU, fp32) for bench.py's second
cpu_baseline leg: the same arithmetic as oracle/ctk_oracle.py:MPPI.step (which follows optimizer_mppi.py:181-193 and is
pinned by the golden vectors), expressed as batched torch ops the way the reference's PyTorch backend runs it, so
that all host cores take part.  Checked against the NumPy oracle in tests/test_oracle_golden.py.  Never imported by the
product (control_toolkit_amd/)."""
import numpy as np
import torch

from . import ctk_oracle as O


class TorchMPPI:
    def __init__(self, mppi: "O.MPPI", threads: int = 0):
        """mppi: a configured NumPy-oracle MPPI (ODE predictor); its constants and nominal plan are taken over."""
        assert mppi.predictor.kind == "ODE" and mppi.predictor.intermediate_steps == 1
        if threads > 0:
            torch.set_num_threads(threads)
        self.threads = torch.get_num_threads()
        self.m = mppi
        self.k = {kk: float(v) for kk, v in O.derived_constants(mppi.predictor.env, mppi.predictor.dt, 1).items()}
        self.env = mppi.predictor.env
        self.M = torch.from_numpy(np.ascontiguousarray(mppi.M[:, :, 0]))        # [P,H]
        self.u_nom = torch.from_numpy(mppi.u_nom.copy()).reshape(-1)            # [H]
        self.u = float(mppi.u)

    @torch.no_grad()
    def step(self, s, noise) -> float:
        m, k, e = self.m, self.k, self.env
        N, H = m.N, m.H
        noise = torch.as_tensor(noise, dtype=torch.float32).reshape(N, -1)
        u_nom = torch.cat([self.u_nom[1:], self.u_nom[-1:]])                    # optimizer_mppi.py:184
        du = (noise * float(m.stdev)) @ self.M                                  # :170-179  [N,H]
        u = torch.clamp(u_nom[None, :] + du, float(m.low), float(m.high))       # :186-187
        x = torch.full((N,), float(s[0])); v = torch.full((N,), float(s[1]))
        th = torch.full((N,), float(s[2])); om = torch.full((N,), float(s[3]))
        inv_xs2dd = e.dd_weight * k["inv_xs"] * k["inv_xs"]
        stage = torch.zeros(N)
        uprev = torch.full((N,), self.u)
        for h in range(H):                                                      # predict_core + stage cost, fused per step
            uh = u[:, h]
            sn, cs = torch.sin(th), torch.cos(th)
            dx = x - e.target_position
            omc = 1.0 - cs
            dut = uh - uprev
            stage += inv_xs2dd * dx * dx + k["ep_c"] * omc * omc + e.ekp_weight * om * om + k["ccR"] * uh * uh \
                + e.ccrc_weight * dut * dut
            A = k["u_max"] * uh + k["k_ml"] * om * om * sn - k["M_fric"] * v
            tmp = A * k["inv_mt"]
            thdd = (k["g"] * sn - cs * tmp - k["k_jf"] * om) / (k["k43l"] - k["k_mpl_mt"] * cs * cs)
            xdd = tmp - k["k_mpl_mt"] * thdd * cs
            x, v, th, om = x + k["dt"] * v, v + k["dt"] * xdd, th + k["dt"] * om, om + k["dt"] * thdd
            uprev = uh
        dx = x - e.target_position
        omc = 1.0 - torch.cos(th)
        terminal = e.terminal_weight * (inv_xs2dd * dx * dx + k["ep_c"] * omc * omc)
        J = (stage + terminal) / (H + 1)                                        # Cost_Functions/__init__.py:90-93
        R, NU = float(m.R), float(m.NU)
        J = J + float(m.cc_weight) * (0.5 * (1.0 - 1.0 / NU) * R * du * du + R * u * du + 0.5 * R * u * u).sum(1)   # :154-155
        w = torch.exp((-1.0 / float(m.LBD)) * (J - J.min()))                    # :163-168
        b = (w[:, None] * du).sum(0) / w.sum()
        self.u_nom = torch.clamp(u_nom + b, float(m.low), float(m.high))        # :190
        self.u = float(self.u_nom[0])                                           # :191
        self.J = J
        return self.u
