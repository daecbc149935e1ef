"""CPU tests of the serving harness: ZMTP framing against the protocol's byte layout, the
reference server loop's request handling (controller_server.py:55-86) and the client behaviour of
controller_remote.py:70-108 (rid matching, stale replies, timeout) over a real TCP loopback."""
import json
import struct
import threading

import numpy as np
import pytest

from control_toolkit_amd.controller_server import zmtp
from control_toolkit_amd.controller_server.controller_server import handle_request, split_parts, serve


class FakeController:
    def __init__(self):
        self.calls = []

    def step(self, s, time=None, updated_attributes={}):
        self.calls.append((np.asarray(s).copy(), time, dict(updated_attributes)))
        if time == "boom":
            raise RuntimeError("controller failed")
        if time == "array":
            return np.array([0.25, -0.5], np.float32)
        return float(np.sum(s)) + float(updated_attributes.get("target_position", 0.0))


def test_zmtp_byte_layout():
    g = zmtp.GREETING
    assert len(g) == 64 and g[0] == 0xFF and g[9] == 0x7F and g[10:12] == b"\x03\x00"
    assert g[12:32] == b"NULL" + b"\x00" * 16 and g[32] == 0 and g[33:] == b"\x00" * 31
    assert zmtp.encode_frame(b"abc") == b"\x00\x03abc"
    assert zmtp.encode_frame(b"abc", more=True) == b"\x01\x03abc"
    big = bytes(300)
    assert zmtp.encode_frame(big) == b"\x02" + struct.pack(">Q", 300) + big
    r = zmtp.ready_command("DEALER")
    assert r[0] == 0x04 and r[1] == len(r) - 2 and r[2:8] == b"\x05READY"
    assert zmtp.parse_ready(r[2:]) == {"socket-type": b"DEALER", "identity": b""}
    # incremental decoding across arbitrary packet boundaries, multi-part messages, commands in between
    stream = zmtp.GREETING + zmtp.ready_command("DEALER") + zmtp.encode_frame(b"", more=True) + zmtp.encode_frame(b"x" * 300) \
        + zmtp.encode_frame(b"\x04PING\x00\x10ctx", command=True) + zmtp.encode_frame(b"tail")

    class Sink:
        def __init__(self): self.sent = b""
        def sendall(self, b): self.sent += b
    for chunk in (1, 7, 64, 1000):
        sink = Sink()
        p = zmtp._Peer(sink)
        msgs = []
        for i in range(0, len(stream), chunk):
            msgs += p.feed(stream[i:i + chunk])
        assert p.ready and msgs == [[b"", b"x" * 300], [b"tail"]]
        assert sink.sent == zmtp.encode_frame(b"\x04PONGctx", command=True)
    with pytest.raises(zmtp.ProtocolError):
        zmtp._Peer(None).feed(b"GET / HTTP/1.1\r\n" + bytes(64))


def test_request_handling_matches_reference_loop():
    c = FakeController()
    rep = json.loads(handle_request(c, json.dumps({"rid": 7, "state": [1, 2, 3, 4], "time": 0.5,
                                                   "updated_attributes": {"target_position": 0.5}}).encode()))
    assert rep == {"rid": 7, "Q": 10.5} and c.calls[0][0].dtype == np.float32 and c.calls[0][1] == 0.5
    rep = json.loads(handle_request(c, json.dumps({"rid": 8, "state": [0, 0, 0, 0], "time": "array"}).encode()))
    assert rep == {"rid": 8, "Q": [0.25, -0.5]}
    assert handle_request(c, json.dumps({"rid": 9, "state": [0, 0, 0, 0], "time": "boom"}).encode()) is None   # no reply (:83-85)
    assert handle_request(c, b"not json") is None
    assert split_parts([b"id", b"p"]) == (b"id", b"p") and split_parts([b"id", b"", b"p"]) == (b"id", b"p")
    assert split_parts([b"id"]) == (None, None) and split_parts([b"id", b"x", b"p"]) == (None, None)


def test_router_dealer_loopback_with_timeouts_and_stale_replies():
    c = FakeController()
    port_box, done = [], threading.Event()
    th = threading.Thread(target=lambda: (serve(c, "127.0.0.1", 0, on_ready=port_box.append, prefer_zmq=False), done.set()), daemon=True)
    th.start()
    while not port_box:
        pass
    d = zmtp.DealerSocket(rcvtimeo_ms=500)
    d.connect("127.0.0.1", port_box[0])
    e = zmtp.DealerSocket(rcvtimeo_ms=500)                      # a second client: replies are routed by identity
    e.connect("127.0.0.1", port_box[0])
    for rid in range(5):
        d.send_json({"rid": rid, "state": [rid, 0, 0, 1], "time": None, "updated_attributes": {}})
        e.send_json({"rid": 100 + rid, "state": [0, 0, 0, 2], "time": None, "updated_attributes": {}})
        assert d.recv_json() == {"rid": rid, "Q": rid + 1.0}
        assert e.recv_json() == {"rid": 100 + rid, "Q": 2.0}
    # controller exception -> no reply -> the client's receive times out (controller_remote.py:84-92)
    d.send_json({"rid": 50, "state": [0, 0, 0, 0], "time": "boom"})
    with pytest.raises(zmtp.Again):
        d.recv_json(timeout_ms=100)
    # stale-reply discipline of controller_remote.py:94-104: skip replies whose rid is not the last request's
    d.send_json({"rid": 51, "state": [1, 1, 1, 1], "time": None})
    d.send_json({"rid": 52, "state": [2, 2, 2, 2], "time": None})
    resp = d.recv_json()
    while resp.get("rid") != 52:
        resp = d.recv_json()
    assert resp == {"rid": 52, "Q": 8.0}
    # a long payload takes the 8-byte size field both ways
    d.send_json({"rid": 53, "state": [0, 0, 0, 0], "time": "array", "updated_attributes": {"pad": "x" * 1000}})
    assert d.recv_json() == {"rid": 53, "Q": [0.25, -0.5]}
    d.send(b"__shutdown__")
    assert done.wait(5)
    d.close(); e.close()
