"""Minimal ZMTP 3.0 (ZeroMQ wire protocol, RFC 23, NULL security mechanism) ROUTER and DEALER endpoints on
plain TCP sockets — just enough for the reference's serving pair: `zmq.ROUTER` in
controller_server/controller_server.py:50-52 and `zmq.DEALER` in Controllers/controller_remote.py:29-33.

pyzmq / libzmq are not part of this image, so this is written from the protocol specification:
  greeting   64 bytes: 0xFF, 8 padding bytes, 0x7F | version 3.0 | mechanism "NULL" padded to 20 | as-server 0 | 31 filler
  handshake  one READY command each way: frame flag 0x04, body = 5 "READY" + properties (1-byte name length, name,
             4-byte big-endian value length, value): Socket-Type = ROUTER | DEALER, Identity = ""
  traffic    frames: flags (0x01 MORE, 0x02 LONG, 0x04 COMMAND), 1-byte or 8-byte big-endian size, body;
             a ROUTER prefixes each received message with the peer's identity and routes sends by it.
It has been exercised against itself only (tests/test_serving_cpu.py); when pyzmq is importable the server
uses the real thing instead (controller_server.py)."""
import selectors
import socket
import struct
import time

GREETING = b"\xff" + b"\x00" * 7 + b"\x01" + b"\x7f" + b"\x03\x00" + b"NULL" + b"\x00" * 16 + b"\x00" + b"\x00" * 31
assert len(GREETING) == 64
FLAG_MORE, FLAG_LONG, FLAG_COMMAND = 0x01, 0x02, 0x04


class Again(Exception):
    """receive timed out (zmq.error.Again)"""


class ProtocolError(Exception):
    pass


def encode_frame(body: bytes, more: bool = False, command: bool = False) -> bytes:
    flags = (FLAG_MORE if more else 0) | (FLAG_COMMAND if command else 0)
    if len(body) > 255:
        return bytes([flags | FLAG_LONG]) + struct.pack(">Q", len(body)) + body
    return bytes([flags, len(body)]) + body


def ready_command(socket_type: str, identity: bytes = b"") -> bytes:
    def prop(name: bytes, value: bytes) -> bytes:
        return bytes([len(name)]) + name + struct.pack(">I", len(value)) + value
    body = b"\x05READY" + prop(b"Socket-Type", socket_type.encode()) + prop(b"Identity", identity)
    return encode_frame(body, command=True)


def parse_ready(body: bytes) -> dict:
    if not body.startswith(b"\x05READY"):
        raise ProtocolError(f"expected READY, got {body[:16]!r}")
    props, i = {}, 6
    while i < len(body):
        n = body[i]; name = body[i + 1:i + 1 + n]; i += 1 + n
        (m,) = struct.unpack(">I", body[i:i + 4]); props[name.decode().lower()] = body[i + 4:i + 4 + m]; i += 4 + m
    return props


class FrameReader:
    """incremental decoder: feed(bytes) -> complete frames as (flags, body)"""

    def __init__(self):
        self.buf = bytearray()

    def feed(self, data: bytes):
        self.buf += data
        out = []
        while True:
            if len(self.buf) < 2:
                break
            flags = self.buf[0]
            if flags & FLAG_LONG:
                if len(self.buf) < 9:
                    break
                (size,) = struct.unpack(">Q", bytes(self.buf[1:9])); head = 9
            else:
                size, head = self.buf[1], 2
            if len(self.buf) < head + size:
                break
            out.append((flags, bytes(self.buf[head:head + size])))
            del self.buf[:head + size]
        return out


class _Peer:
    def __init__(self, sock, identity=None):
        self.sock, self.identity = sock, identity
        self.greeting = bytearray()
        self.reader = FrameReader()
        self.ready = False
        self.parts = []          # frames of the message being assembled
        self.props = {}

    def feed(self, data: bytes):
        """-> list of complete messages (each a list of frame bodies)"""
        if len(self.greeting) < 64:
            need = 64 - len(self.greeting)
            self.greeting += data[:need]
            data = data[need:]
            if len(self.greeting) == 64:
                if self.greeting[0] != 0xFF or self.greeting[9] != 0x7F or self.greeting[10] < 3:
                    raise ProtocolError("not a ZMTP 3.x greeting")
                if not bytes(self.greeting[12:32]).startswith(b"NULL"):
                    raise ProtocolError("only the NULL security mechanism is implemented")
            if not data:
                return []
        msgs = []
        for flags, body in self.reader.feed(data):
            if flags & FLAG_COMMAND:
                if not self.ready:
                    self.props = parse_ready(body)
                    self.ready = True
                elif body.startswith(b"\x04PING"):              # ZMTP 3.1 heartbeat: answer with the context echoed
                    self.sock.sendall(encode_frame(b"\x04PONG" + body[7:], command=True))
                continue
            self.parts.append(body)
            if not flags & FLAG_MORE:
                msgs.append(self.parts); self.parts = []
        return msgs


class RouterSocket:
    """zmq.ROUTER: recv_multipart() -> [identity, *frames]; send_multipart([identity, *frames])."""

    def __init__(self):
        self.sel = selectors.DefaultSelector()
        self.listener = None
        self.peers = {}          # identity -> _Peer
        self.queue = []
        self._next_id = 1

    def bind(self, host: str = "0.0.0.0", port: int = 5555) -> int:
        self.listener = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.listener.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.listener.bind((host, port))
        self.listener.listen(16)
        self.listener.setblocking(False)
        self.sel.register(self.listener, selectors.EVENT_READ, None)
        return self.listener.getsockname()[1]

    def _accept(self):
        conn, _ = self.listener.accept()
        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        conn.sendall(GREETING + ready_command("ROUTER"))
        conn.setblocking(False)
        ident = b"\x00" + struct.pack(">I", self._next_id); self._next_id += 1     # libzmq-style generated identity
        peer = _Peer(conn, ident)
        self.peers[ident] = peer
        self.sel.register(conn, selectors.EVENT_READ, peer)

    def _drop(self, peer):
        try:
            self.sel.unregister(peer.sock)
        except Exception:
            pass
        peer.sock.close()
        self.peers.pop(peer.identity, None)

    def recv_multipart(self, timeout_s=None):
        deadline = None if timeout_s is None else time.monotonic() + timeout_s
        while not self.queue:
            left = None if deadline is None else max(0.0, deadline - time.monotonic())
            events = self.sel.select(left)
            if not events and deadline is not None and time.monotonic() >= deadline:
                raise Again()
            for key, _ in events:
                if key.data is None:
                    self._accept()
                    continue
                peer = key.data
                try:
                    data = peer.sock.recv(65536)
                    if not data:
                        self._drop(peer); continue
                    for msg in peer.feed(data):
                        if peer.props.get("identity"):           # a peer that names itself keeps its name
                            new = bytes(peer.props["identity"])
                            if new != peer.identity and new not in self.peers:
                                self.peers.pop(peer.identity, None); peer.identity = new; self.peers[new] = peer
                        self.queue.append([peer.identity] + msg)
                except (ProtocolError, ConnectionError, OSError):
                    self._drop(peer)
        return self.queue.pop(0)

    def send_multipart(self, parts):
        peer = self.peers.get(bytes(parts[0]))
        if peer is None:
            return                                               # ROUTER drops messages for unknown peers
        frames = parts[1:]
        data = b"".join(encode_frame(bytes(f), more=(i + 1 < len(frames))) for i, f in enumerate(frames))
        try:
            peer.sock.setblocking(True); peer.sock.sendall(data); peer.sock.setblocking(False)
        except OSError:
            self._drop(peer)

    def close(self):
        for p in list(self.peers.values()):
            self._drop(p)
        if self.listener is not None:
            self.sel.unregister(self.listener); self.listener.close(); self.listener = None


class DealerSocket:
    """zmq.DEALER client side of controller_remote.py:29-33,70-108: send_json / recv_json with RCVTIMEO."""

    def __init__(self, rcvtimeo_ms=None):
        self.sock = None
        self.peer = None
        self.rcvtimeo_ms = rcvtimeo_ms
        self.inbox = []

    def connect(self, host: str, port: int, timeout_s: float = 10.0):
        self.sock = socket.create_connection((host, port), timeout=timeout_s)
        self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        self.sock.sendall(GREETING + ready_command("DEALER"))
        self.peer = _Peer(self.sock)
        t0 = time.monotonic()
        while not self.peer.ready:                               # wait for the server's greeting + READY
            if time.monotonic() - t0 > timeout_s:
                raise ProtocolError("no ZMTP handshake from the server")
            self.inbox += self.peer.feed(self.sock.recv(65536))
        if self.peer.props.get("socket-type") != b"ROUTER":
            raise ProtocolError(f"peer is {self.peer.props.get('socket-type')!r}, expected ROUTER")

    def send(self, payload: bytes):
        self.sock.sendall(encode_frame(payload))

    def recv(self, timeout_ms=None) -> bytes:
        timeout_ms = self.rcvtimeo_ms if timeout_ms is None else timeout_ms
        deadline = None if timeout_ms is None else time.monotonic() + timeout_ms / 1e3
        while not self.inbox:
            left = None if deadline is None else deadline - time.monotonic()
            if left is not None and left <= 0:
                raise Again()
            self.sock.settimeout(left)
            try:
                data = self.sock.recv(65536)
            except socket.timeout:
                raise Again() from None
            if not data:
                raise ConnectionError("server closed the connection")
            self.inbox += self.peer.feed(data)
        return self.inbox.pop(0)[-1]

    def send_json(self, obj):
        import json
        self.send(json.dumps(obj).encode("utf-8"))

    def recv_json(self, timeout_ms=None):
        import json
        return json.loads(self.recv(timeout_ms).decode("utf-8"))

    def close(self):
        if self.sock is not None:
            self.sock.close(); self.sock = None
