"""CPU, build container only (needs /root/reference; skipped on the GPU box).  DESIGN.md section 3's one semantic choice —
`lib.assign` of the un-vendored SI_Toolkit as VALUE (TensorFlow) semantics — is MEASURED here, not argued: the unmodified
reference optimizer_rpgd runs twice on the same seeds and states, once per semantics (tests/ref_assign_probe.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists in the build container only")


@pytest.mark.timeout(180)
def test_inplace_assign_changes_only_the_applied_control_and_makes_it_a_fresh_sample():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_assign_probe.py")], capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("ASSIGN_JSON ")][-1][len("ASSIGN_JSON "):])
    assert d["value_run_equals_fixture"]                        # the probe's value-semantics run IS the committed fixture's run
    diff = d["max_abs_diff"]
    # step 0, identical inputs: the population, the Adam moments and the ages do not depend on the semantics ...
    assert diff["Q_0"] == 0.0 and diff["m_0"] == 0.0 and diff["v_0"] == 0.0 and diff["ages_0"] == 0.0
    # ... the control that is applied does, and grossly: with an in-place assign `u_nom = Q_tf[None, best_idx[0]]` (a view,
    # optimizer_rpgd.py:426) follows the warm-started population written at :515, so u = first input of ROW best_idx[0] of the NEW
    # population — a freshly resampled plan on a resampling step — instead of the best plan's
    assert diff["u_0"] > 0.1 and diff["u_nom_0"] > 0.1
    assert all(d["inplace_u_is_row_of_new_population"])
    # later steps: the populations drift apart only through that control (it is the next step's previous input in the cost)
    for t in range(1, d["steps"]):
        assert diff[f"ages_{t}"] == 0.0 and diff[f"Q_{t}"] < 5e-3 and diff[f"u_{t}"] > 1e-3
