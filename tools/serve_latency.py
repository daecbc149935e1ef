#!/usr/bin/env python3
"""External-process latency of the drop-in: a server PROCESS runs controller_mpc with the HIP
MPPI optimizer behind the reference's ROUTER loop; this process plays Controllers/controller_remote.py (DEALER,
request ids, closed loop with a host plant) and reports the round-trip distribution against the reference
client's 50 ms receive deadline (controller_remote.py:11).  usage: python tools/serve_latency.py [--requests 2000]"""
import argparse, json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_toolkit_amd.controller_server.zmtp import DealerSocket, Again   # noqa: E402
from bench import plant_step                                                  # noqa: E402  (host plant, bench plumbing)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--requests", type=int, default=2000)
    ap.add_argument("--port", type=int, default=5617)
    ap.add_argument("--optimizer", default="mppi-hip")
    args = ap.parse_args()
    srv = subprocess.Popen([sys.executable, "-m", "control_toolkit_amd.controller_server.controller_server",
                            "--optimizer", args.optimizer, "--host", "127.0.0.1", "--port", str(args.port)],
                           cwd=ROOT, stderr=subprocess.PIPE, text=True)
    try:
        line = ""
        t0 = time.time()
        while "listening" not in line:                      # controller built, engine created, socket bound
            line = srv.stderr.readline()
            if not line and srv.poll() is not None:
                raise SystemExit("server died during start-up")
            if time.time() - t0 > 300:
                raise SystemExit("server start-up timed out")
        transport = line.strip().split("(")[-1].rstrip(")")
        d = DealerSocket(rcvtimeo_ms=50)                    # DEFAULT_RCVTIMEO of the reference client
        d.connect("127.0.0.1", args.port)
        s = np.array([0.0, 0.0, 0.2, 0.0], np.float32)
        rtt, late = [], 0
        for rid in range(args.requests + 50):
            t1 = time.perf_counter()
            d.send_json({"rid": rid, "state": s.tolist(), "time": rid * 0.02, "updated_attributes": {"target_position": 0.0}})
            try:
                resp = d.recv_json()
                while resp.get("rid") != rid:
                    resp = d.recv_json()
            except Again:
                late += 1
                continue
            dt = time.perf_counter() - t1
            if rid >= 50:
                rtt.append(dt)
            plant_step(s, float(np.asarray(resp["Q"]).reshape(-1)[0]))
        d.send(b"__shutdown__")
        r = np.array(rtt) * 1e6
        print(json.dumps({"what": "external-process round trip: DEALER client -> ROUTER server -> controller_mpc.step (mppi-hip, "
                                  "N=1024, H=50) -> reply", "transport": transport, "requests": len(rtt), "missed_50ms_deadline": late,
                          "rtt_us_median": float(np.median(r)), "rtt_us_p95": float(np.percentile(r, 95)),
                          "rtt_us_p99": float(np.percentile(r, 99)), "rtt_us_max": float(r.max()),
                          "deadline_us": 50000, "headroom_x_at_p99": 50000 / float(np.percentile(r, 99))}))
    finally:
        try:
            srv.wait(timeout=10)
        except Exception:
            srv.kill()


if __name__ == "__main__":
    main()
