"""-m gpu: the serving harness end to end on the GPU — the reference's ROUTER loop
(controller_server/controller_server.py:55-86) around controller_mpc + `mppi-hip`, a DEALER client behaving like
controller_remote.py:70-108 (request ids, 50 ms receive deadline), closed loop against the oracle plant.
Transport: the in-tree ZMTP 3.0 endpoints (pyzmq is not in this image; interoperability with libzmq is untested)."""
import threading

import numpy as np
import pytest

from oracle import ctk_oracle as O
from control_toolkit_amd.controller_server import zmtp
from control_toolkit_amd.controller_server.controller_server import build_controller, serve

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(120)
def test_router_loop_serves_mppi_hip_within_the_reference_deadline():
    cfg = dict(num_rollouts=3500, mpc_horizon=35, period_interpolation_inducing_points=10, seed=5)   # the reference's default MPPI size
    ctrl = build_controller("mppi-hip", "ODE", cfg)
    local = build_controller("mppi-hip", "ODE", cfg)            # same seed (same device draws), stepped in-process
    port_box, done = [], threading.Event()
    th = threading.Thread(target=lambda: (serve(ctrl, "127.0.0.1", 0, on_ready=port_box.append, prefer_zmq=False), done.set()), daemon=True)
    th.start()
    while not port_box:
        pass
    d = zmtp.DealerSocket(rcvtimeo_ms=50)                       # controller_remote.py:11: 50 ms receive timeout
    d.connect("127.0.0.1", port_box[0])
    pred = O.Predictor("ODE")
    s = np.array([0.0, 0.0, 0.15, 0.0], np.float32)
    missed = 0
    for rid in range(60):
        upd = {"target_position": 0.05} if rid == 30 else {}
        d.send_json({"rid": rid, "state": [float(x) for x in s], "time": 0.02 * rid, "updated_attributes": upd})
        try:
            rep = d.recv_json()
        except zmtp.Again:
            missed += 1
            continue
        assert rep["rid"] == rid and isinstance(rep["Q"], float) and abs(rep["Q"]) <= 1.0
        u_local = float(np.asarray(local.step(s, 0.02 * rid, upd)).reshape(-1)[0])
        assert rep["Q"] == pytest.approx(u_local, abs=1e-6)       # the served controller IS controller_mpc.step on the GPU
        s = pred.step(s.reshape(1, 4), np.array([rep["Q"]], np.float32))[0]
    assert missed == 0, f"{missed} replies missed the 50 ms deadline"
    assert abs(s[2]) < 0.3, f"pole fell: angle {s[2]}"
    assert ctrl.optimizer.engine.get_param("target_position") == np.float32(0.05)   # updated_attributes reached the kernels
    # a malformed state makes controller.step raise: logged, NO reply (controller_server.py:83-85); the loop keeps serving
    d.send_json({"rid": 100, "state": [0.0, 0.0], "time": None})
    with pytest.raises(zmtp.Again):
        d.recv_json(timeout_ms=100)
    d.send_json({"rid": 101, "state": [0.0, 0.0, 0.1, 0.0], "time": None})
    assert d.recv_json(timeout_ms=500)["rid"] == 101
    d.send(b"__shutdown__")
    assert done.wait(5)
    d.close()
