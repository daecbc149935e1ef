#!/usr/bin/env python3
"""controller_server — mirror of reference controller_server/controller_server.py:30-86 without the GUI picker:
a ROUTER endpoint that serves `{"rid","state","time","updated_attributes"}` -> `{"rid","Q"}` (README.md:127-137
of the reference) by calling `ctrl.step`, one request at a time (a handle is single-threaded, as the reference's
loop is).  Exceptions in the controller: logged, NO reply (reference :83-85) — the client's 50 ms timeout
(controller_remote.py:11) is the error path.

Transport: pyzmq's ROUTER socket when importable, otherwise the in-tree ZMTP 3.0 endpoint (zmtp.py)."""
import json
import sys

import numpy as np


def handle_request(ctrl, payload: bytes):
    """reference :70-82.  Returns the reply payload, or None when nothing is to be sent back."""
    try:
        req = json.loads(payload.decode("utf-8"))
        rid = req["rid"]
        s = np.asarray(req["state"], dtype=np.float32)
        t = req.get("time")
        upd = req.get("updated_attributes", {}) or {}
        Q = ctrl.step(s, t, upd)
        if isinstance(Q, np.ndarray):
            Q_payload = Q.tolist()
        else:
            Q_payload = float(Q) if not isinstance(Q, (list, tuple)) else Q
        return json.dumps({"rid": rid, "Q": Q_payload}).encode("utf-8")
    except Exception as e:   # noqa: BLE001 — reference :83-85
        print(f"[server] controller exception - no reply sent: {e}", file=sys.stderr)
        return None


def split_parts(parts):
    """reference :60-68: [identity, payload] or [identity, b"", payload]; anything else is skipped"""
    if len(parts) == 2:
        return parts[0], parts[1]
    if len(parts) == 3 and parts[1] == b"":
        return parts[0], parts[2]
    return None, None


def open_router(host: str, port: int, prefer_zmq: bool = True):
    """-> (recv_multipart, send_multipart, close, bound_port, transport_name)"""
    if prefer_zmq:
        try:
            import zmq                                           # the reference's transport (:49-52)
            ctx = zmq.Context()
            sock = ctx.socket(zmq.ROUTER)
            if port == 0:
                port = sock.bind_to_random_port(f"tcp://{host}")
            else:
                sock.bind(f"tcp://{host}:{port}")
            return sock.recv_multipart, sock.send_multipart, lambda: (sock.close(0), ctx.term()), port, "pyzmq"
        except ImportError:
            pass
    from .zmtp import RouterSocket
    r = RouterSocket()
    port = r.bind(host, port)
    return r.recv_multipart, r.send_multipart, r.close, port, "zmtp (in-tree)"


def serve(ctrl, host: str = "0.0.0.0", port: int = 5555, max_requests=None, on_ready=None, prefer_zmq: bool = True):
    recv, send, close, port, transport = open_router(host, port, prefer_zmq)
    print(f"[server] listening on tcp://{host}:{port} ({transport})", file=sys.stderr)
    if on_ready is not None:
        on_ready(port)
    served = 0
    try:
        while max_requests is None or served < max_requests:
            ident, payload = split_parts(recv())
            if ident is None:
                continue
            if payload == b"__shutdown__":                       # harness convenience, not part of the reference protocol
                break
            reply = handle_request(ctrl, payload)
            served += 1
            if reply is not None:
                send([ident, reply])
    finally:
        close()
    return served


def build_controller(optimizer: str = "mppi-hip", predictor_specification: str = "ODE", config_optimizer: dict = None,
                     device: str = "gpu:0"):
    """the reference builds its controller from the GUI's choice (:32-47); here from arguments"""
    from ..Controllers.controller_mpc import controller_mpc
    defaults = {"mppi-hip": dict(seed=1, mpc_horizon=50, num_rollouts=1024, cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0,
                                 SQRTRHOINV=0.03, period_interpolation_inducing_points=1, mpc_timestep=0.02)}
    cfg = dict(defaults.get(optimizer, {}), **(config_optimizer or {}))
    lim = (np.array([-1.0], np.float32), np.array([1.0], np.float32))
    c = controller_mpc("CartPole", lim, {"target_position": 0.0, "target_equilibrium": 1.0},
                       config_controllers={"mpc": {"optimizer": optimizer, "predictor_specification": predictor_specification,
                                                   "computation_library": "hip", "controller_logging": False,
                                                   "calculate_optimal_trajectory": False, "device": device}},
                       config_optimizers={optimizer: cfg})
    c.configure()
    return c


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--optimizer", default="mppi-hip")
    ap.add_argument("--predictor", default="ODE")
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=5555)          # reference ENDPOINT tcp://*:5555 (:19)
    ap.add_argument("--max-requests", type=int, default=None)
    args = ap.parse_args(argv)
    ctrl = build_controller(args.optimizer, args.predictor)
    serve(ctrl, args.host, args.port, args.max_requests)


if __name__ == "__main__":
    main()
