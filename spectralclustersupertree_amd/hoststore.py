"""Host-side rendezvous of the ranks of one job without torch: a tiny TCP star.

The data path of a multi-GPU run is RCCL inside libscs_hip.so; what the HOST side needs is
small and rare -- ship rank 0's 128-byte RCCL id to everybody once, a barrier, a maximum over
ranks for the benchmark's clock, the exchange of a few subtrees in "forked" teams.  North star:
"no PyTorch"; ``torch.distributed.run`` stays a LAUNCHER only (it sets RANK, WORLD_SIZE,
LOCAL_RANK, MASTER_ADDR, MASTER_PORT and starts one process per GPU).

Rank 0 listens, every other rank connects; every collective is "send to rank 0, rank 0
answers everybody" (messages are tens of bytes to a few kilobytes, the ranks live on one
node).  Port: ``SCS_RDZV_PORT`` or MASTER_PORT + 1 ... + 16 -- the launcher's own store owns
MASTER_PORT itself -- rank 0 takes the first one it can bind, the others probe the same list and
recognise the job by a handshake (magic, world size, job tag), so a stranger listening on one of
the ports is skipped.
"""

from __future__ import annotations

import os
import pickle
import socket
import struct
import time

_MAGIC = b"SCSRDZV1"
_PROBES = 16

# What travels is plain data -- None, numbers, strings, bytes (the RCCL id), tuples / lists /
# dicts of those (flat trees), numpy scalars and arrays.  The sockets are unauthenticated (the
# handshake only tells jobs apart), so nothing received is allowed to name any other global: a
# pickle that asks for one is refused instead of executed.
_SAFE_GLOBALS = {
    ("builtins", "complex"), ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "bytearray"),
    ("builtins", "slice"), ("builtins", "range"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
}


class _DataUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _SAFE_GLOBALS:
            return super().find_class(module, name)
        msg = f"rendezvous: refused to unpickle {module}.{name} (only plain data travels between the ranks)"
        raise pickle.UnpicklingError(msg)


def _loads(blob: bytes):
    import io

    return _DataUnpickler(io.BytesIO(blob)).load()


def _send(sock: socket.socket, payload: bytes) -> None:
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    chunks = []
    while n:
        chunk = sock.recv(min(n, 1 << 20))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous connection")
        chunks.append(chunk)
        n -= len(chunk)
    return b"".join(chunks)


def _recv(sock: socket.socket) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class HostGroup:
    """The ranks of one job on the host side: ``allgather`` (any picklable object),
    ``broadcast`` from rank 0, ``barrier``, ``max``.  Every rank makes the same calls in the
    same order."""

    def __init__(self, rank: int, world: int, addr: str | None = None, port: int | None = None,
                 timeout: float = 120.0, tag: str | None = None) -> None:
        self.rank, self.world = rank, world
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        if port is None:
            port = int(os.environ.get("SCS_RDZV_PORT", "0")) or int(os.environ.get("MASTER_PORT", "29500")) + 1
        tag = tag if tag is not None else os.environ.get("TORCHELASTIC_RUN_ID", "")
        self._hello = _MAGIC + struct.pack("<I", world) + tag.encode()[:64]
        self._peers: list[socket.socket | None] = [None] * world  # rank 0 only
        self._sock: socket.socket | None = None  # ranks > 0
        deadline = time.monotonic() + timeout
        if world <= 1:
            return
        if rank == 0:
            self._serve(addr, port, deadline)
        else:
            self._join(addr, port, deadline)

    # ---- set-up ---------------------------------------------------------------------------
    def _serve(self, addr: str, port: int, deadline: float) -> None:
        listener = None
        for p in range(port, port + _PROBES):
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((addr, p))
                s.listen(self.world + 8)
                listener = s
                break
            except OSError:
                s.close()
        if listener is None:
            msg = f"rendezvous: cannot bind any of the ports {port}..{port + _PROBES - 1} on {addr}"
            raise OSError(msg)
        joined = 0
        try:
            while joined < self.world - 1:
                listener.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    conn, _ = listener.accept()
                except socket.timeout:
                    msg = f"rendezvous: {self.world - 1 - joined} rank(s) did not join in time"
                    raise TimeoutError(msg) from None
                conn.settimeout(10.0)
                try:
                    hello = _recv(conn)
                    if not hello.startswith(self._hello) or len(hello) != len(self._hello) + 4:
                        conn.close()  # a stranger (or another job): not ours
                        continue
                    (peer,) = struct.unpack("<I", hello[-4:])
                    if not 0 < peer < self.world or self._peers[peer] is not None:
                        conn.close()
                        continue
                    _send(conn, self._hello)
                except (OSError, struct.error, ConnectionError):
                    conn.close()
                    continue
                conn.settimeout(None)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self._peers[peer] = conn
                joined += 1
        finally:
            listener.close()

    def _join(self, addr: str, port: int, deadline: float) -> None:
        while True:
            for p in range(port, port + _PROBES):
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.settimeout(2.0)
                try:
                    s.connect((addr, p))
                    _send(s, self._hello + struct.pack("<I", self.rank))
                    if _recv(s) == self._hello:
                        s.settimeout(None)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self._sock = s
                        return
                except (OSError, struct.error, ConnectionError):
                    pass
                s.close()
            if time.monotonic() > deadline:
                msg = f"rendezvous: rank {self.rank} found no rank 0 on {addr}:{port}..{port + _PROBES - 1}"
                raise TimeoutError(msg)
            time.sleep(0.05)

    # ---- collectives ----------------------------------------------------------------------
    def allgather(self, obj) -> list:
        if self.world <= 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [None] * (self.world - 1)
            for r in range(1, self.world):
                out[r] = _loads(_recv(self._peers[r]))
            blob = pickle.dumps(out, protocol=pickle.HIGHEST_PROTOCOL)
            for r in range(1, self.world):
                _send(self._peers[r], blob)
            return out
        _send(self._sock, pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL))
        return _loads(_recv(self._sock))

    def broadcast(self, obj):
        """Rank 0's object on every rank."""
        return self.allgather(obj if self.rank == 0 else None)[0]

    def barrier(self) -> None:
        self.allgather(None)

    def max(self, value: float) -> float:
        return max(self.allgather(float(value)))

    def close(self) -> None:
        for s in [self._sock, *self._peers]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._sock = None
        self._peers = [None] * self.world
