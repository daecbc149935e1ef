"""CPU, build container only (needs /root/reference; skipped on the GPU box): the unmodified reference controller_mpc
resolves and drives the `*-hip` plug-ins (VERDICT r1 item 6 / INTEGRATION.md section 3: "zero edits to the caller")."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists in the build container only")


@pytest.mark.timeout(120)
def test_unmodified_reference_controller_mpc_drives_the_hip_plugins():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_plugin_probe.py")], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("PROBE_JSON ")][-1]
    d = json.loads(line[len("PROBE_JSON "):])
    # resolved by the reference's own discovery, from the one-line shim in Control_Toolkit_ASF/Optimizers/
    assert d["optimizer_class"] == "optimizer_mppi_hip" and d["optimizer_module"].endswith("optimizer_mppi_hip")
    assert d["lib"] == "NumpyLibrary"                                            # computation_library: numpy — no edit to the library switch
    assert d["predictor_class"].startswith("SI_Toolkit.") and d["cost_class"].startswith("Control_Toolkit.Cost_Functions")   # reference-shaped objects
    kind, optimizer, predictor, kw = d["calls"][0]
    assert (kind, optimizer, predictor) == ("create", "mppi", "ODE")
    assert kw["num_rollouts"] == 96 and kw["mpc_horizon"] == 25 and kw["dt"] == 0.02 and kw["seed"] == 7
    assert kw["action_low"] == [-1.0] and kw["action_high"] == [1.0] and kw["environment"] == "CartPole"
    assert kw["period_interpolation_inducing_points"] == 5 and kw["LBD"] == 100.0 and kw["intermediate_steps"] == 2
    assert d["calls"][1][0] == "reset"                                           # configure() ends with optimizer_reset (optimizer_mppi.py:139)
    steps = [c for c in d["calls"] if c[0] == "step"]
    assert len(steps) == 2 and steps[0][1] == pytest.approx([0.0, 0.1, 0.2, 0.3]) and steps[0][3] == [0.0] and steps[1][3] == [0.25]
    assert d["u"] == 0.25                                                        # squeezed scalar for C = 1 (optimizer_mppi.py:212)
    p = d["params_after_update"]
    assert p["dd_weight"] == 123.0 and p["ep_weight"] == 4567.0                  # Control_Toolkit_ASF/config_cost_function.yml[CartPole][default]
    assert p["m_pole"] == pytest.approx(0.1) and p["target_position"] == pytest.approx(-0.3)   # YAML override; per-step attribute
    assert d["calls"][-1][0] == "reset"                                          # controller_reset -> optimizer_reset
    assert d["rpgd_class"] == "optimizer_rpgd_hip"
    kinds = [c[0] for c in d["rpgd_calls"]]
    assert kinds[0] == "create" and "reset" in kinds and kinds[-1] == "step"
    kw = d["rpgd_calls"][0][3]
    assert kw["opt_keep_k"] == 4 and kw["sample_whole_control_space"] == 1 and kw["outer_its"] == 2
