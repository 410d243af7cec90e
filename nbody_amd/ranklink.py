"""ranklink -- rendezvous of the ranks of ONE node over a Unix-domain socket hub (stdlib only).

bench.py with N > 1 is one Python process per GPU.  Importing torch there would load the HIP runtime and the librccl
bundled with the torch wheel before libnbody_hip.so binds its own (/opt/rocm's), so the few host-side collectives the
harness needs -- barrier, max/min/sum of some floats, a broadcast of RCCL's 128-byte unique id, an all-gather of byte
rows for the host transport -- go through this instead: rank 0 listens on a Unix socket named after the launcher's
MASTER_PORT and run id, the other ranks connect, and every collective is "send mine to rank 0, get everybody's back".
The data path (the per-step all-gather of source positions) is RCCL inside the library and never passes through here.
The C harness (nbody-bench --gpus P) does the same over a shared page (csrc/rank_page.c).

Who may connect, and what may be sent:
  * the socket is a FILE inside a directory this user owns with mode 0700 (`$TMPDIR/nbody_ranklink_<uid>/`): other users
    cannot reach it, and two runs with the same name collide loudly at bind() instead of cross-connecting;
  * both ends check SO_PEERCRED: the peer's uid must be ours;
  * every peer answers the hub's hello with sha256(name | hub pid | nonce | rank) before it is admitted -- a stray
    process of the same user that connects by accident is refused: it gets HELLO_TIMEOUT_S to answer, and silence, garbage
    or a wrong token closes THAT connection only (the hub keeps accepting until its own deadline);
  * the wire format is fixed (a one-byte tag + lengths; None, bytes, int64, a list of float64, one level of list): nothing
    received is ever executed or unpickled.
"""
import hashlib
import os
import socket
import stat
import struct
import tempfile
import time

_MAX_MESSAGE = 1 << 32      # bytes; the largest real message is a few MiB of particle slices
_MAX_ITEMS = 1 << 16
HELLO_TIMEOUT_S = 5.0       # what one connecting peer gets to answer the hub's hello


def encode(obj, _depth=0):
    """None | bytes | int | list of floats | list of those (one level), as tagged bytes."""
    if obj is None:
        return b"N"
    if isinstance(obj, (bytes, bytearray, memoryview)):
        b = bytes(obj)
        return b"B" + struct.pack("<Q", len(b)) + b
    if isinstance(obj, bool):
        raise TypeError("ranklink carries None, bytes, int, float lists and lists of those")
    if isinstance(obj, int):
        return b"I" + struct.pack("<q", obj)
    if isinstance(obj, (list, tuple)):
        if len(obj) > _MAX_ITEMS:
            raise ValueError("ranklink list too long")       # the receiver would refuse it: fail at the sender
        if all(isinstance(v, float) for v in obj) and len(obj) > 0:
            return b"F" + struct.pack("<I", len(obj)) + struct.pack("<%dd" % len(obj), *obj)
        if _depth >= 1:
            raise TypeError("ranklink lists nest one level only")
        return b"L" + struct.pack("<I", len(obj)) + b"".join(encode(v, _depth + 1) for v in obj)
    raise TypeError("ranklink carries None, bytes, int, float lists and lists of those, not %s" % type(obj).__name__)


def decode(buf, pos=0, _depth=0):
    """(object, next position); raises ValueError on anything that is not the format above."""
    try:
        tag = buf[pos:pos + 1]
        pos += 1
        if tag == b"N":
            return None, pos
        if tag == b"B":
            (n,) = struct.unpack_from("<Q", buf, pos)
            pos += 8
            if n > len(buf) - pos:
                raise ValueError("ranklink: truncated bytes")
            return bytes(buf[pos:pos + n]), pos + n
        if tag == b"I":
            (v,) = struct.unpack_from("<q", buf, pos)
            return int(v), pos + 8
        if tag == b"F":
            (n,) = struct.unpack_from("<I", buf, pos)
            pos += 4
            if n > _MAX_ITEMS or 8 * n > len(buf) - pos:
                raise ValueError("ranklink: bad float list")
            return [float(v) for v in struct.unpack_from("<%dd" % n, buf, pos)], pos + 8 * n
        if tag == b"L" and _depth == 0:
            (n,) = struct.unpack_from("<I", buf, pos)
            pos += 4
            if n > _MAX_ITEMS:
                raise ValueError("ranklink: bad list")
            out = []
            for _ in range(n):
                v, pos = decode(buf, pos, 1)
                out.append(v)
            return out, pos
    except struct.error as e:
        raise ValueError("ranklink: truncated message") from e
    raise ValueError("ranklink: unknown tag %r" % tag)


def socket_dir():
    """`$TMPDIR/nbody_ranklink_<uid>`: ours, a real directory, mode 0700 -- or an error."""
    d = os.path.join(tempfile.gettempdir(), "nbody_ranklink_%d" % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError("%s must be a directory owned by uid %d with mode 0700" % (d, os.getuid()))
    return d


def _peer_uid(sock):
    pid, uid, _gid = struct.unpack("3i", sock.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i")))
    return uid, pid


def _token(name, hub_pid, nonce, rank):
    return hashlib.sha256(b"%s|%d|%s|%d" % (name.encode(), hub_pid, nonce, rank)).digest()


class RankLink:
    def __init__(self, rank, world, name=None, timeout_s=900.0):
        self.rank, self.world, self.timeout_s = rank, world, timeout_s
        self.peers, self.hub, self.path = {}, None, None
        if world == 1:
            return
        if name is None:
            name = "nbody_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"))
        name = "".join(c if c.isalnum() or c in "-_." else "_" for c in name)[:80]
        path = os.path.join(socket_dir(), name + ".sock")
        deadline = time.monotonic() + timeout_s
        if rank == 0:
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            try:
                srv.bind(path)   # EADDRINUSE when another run of the same name is alive: loud, not cross-connected
            except OSError:
                # a socket file nobody listens on is the leftover of a run that was killed: replace it
                probe = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    probe.connect(path)
                    probe.close()
                    raise
                except (ConnectionRefusedError, FileNotFoundError):
                    probe.close()
                    os.unlink(path)
                    srv.bind(path)
            self.path = path
            srv.listen(world)
            nonce = os.urandom(16)
            try:
                while len(self.peers) < world - 1:
                    left = deadline - time.monotonic()
                    if left <= 0:
                        raise TimeoutError(f"rank 0: {world - 1 - len(self.peers)} of {world - 1} peers never reached the hub "
                                           f"at {path!r} within {timeout_s} s")
                    srv.settimeout(left)
                    try:
                        conn, _ = srv.accept()
                    except socket.timeout:
                        continue          # the loop's own deadline check words the error
                    # one connection's hello must not hold the hub: a short per-connection bound, and anything that is not
                    # a well-formed answer from a peer of this run closes that connection only
                    try:
                        conn.settimeout(min(HELLO_TIMEOUT_S, max(left, 0.1)))
                        if _peer_uid(conn)[0] != os.getuid():
                            raise ConnectionError("peer of another uid")
                        self._send(conn, nonce)
                        hello = self._recv(conn)
                        ok = (isinstance(hello, list) and len(hello) == 2 and isinstance(hello[0], int) and 0 < hello[0] < world
                              and hello[0] not in self.peers and hello[1] == _token(name, os.getpid(), nonce, hello[0]))
                        if not ok:
                            raise ConnectionError("not a peer of this run")
                    except (ValueError, OSError):   # ConnectionError and socket.timeout are OSErrors; decode() raises ValueError
                        conn.close()
                        continue
                    conn.settimeout(timeout_s)
                    self.peers[hello[0]] = conn
            finally:
                srv.close()
                try:
                    os.unlink(path)   # everybody is connected: the name is free again
                except OSError:
                    pass
                self.path = None
        else:
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(path)
                    break
                except (ConnectionRefusedError, FileNotFoundError):
                    s.close()
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"rank {rank}: no hub at {path!r} within {timeout_s} s")
                    time.sleep(0.02)
            s.settimeout(timeout_s)
            uid, hub_pid = _peer_uid(s)
            if uid != os.getuid():
                s.close()
                raise PermissionError("rank link hub belongs to uid %d" % uid)
            nonce = self._recv(s)
            if not isinstance(nonce, bytes) or len(nonce) != 16:
                s.close()
                raise ConnectionError("rank link hub sent no nonce")
            self.hub = s
            self._send(s, [rank, _token(name, hub_pid, nonce, rank)])

    @staticmethod
    def _send(sock, obj):
        data = encode(obj)
        sock.sendall(struct.pack("<Q", len(data)) + data)

    @staticmethod
    def _recv(sock):
        def exactly(n):
            chunks = []
            while n:
                c = sock.recv(min(n, 1 << 20))
                if not c:
                    raise ConnectionError("peer closed the rank link")
                chunks.append(c)
                n -= len(c)
            return b"".join(chunks)
        (n,) = struct.unpack("<Q", exactly(8))
        if n > _MAX_MESSAGE:
            raise ConnectionError("rank link message of %d bytes" % n)
        data = exactly(n)
        obj, end = decode(data)
        if end != len(data):
            raise ConnectionError("rank link message with trailing bytes")
        return obj

    def allgather(self, obj):
        """Everybody's object, indexed by rank.  The one primitive; the rest is built on it."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            objs = [obj] + [self._recv(self.peers[r]) for r in range(1, self.world)]
            for r in range(1, self.world):
                self._send(self.peers[r], objs)
            return objs
        self._send(self.hub, obj)
        return self._recv(self.hub)

    def barrier(self):
        self.allgather(None)

    def broadcast(self, obj, src=0):
        return self.allgather(obj if self.rank == src else None)[src]

    def reduce(self, values, op):
        """Element-wise max / min / sum of a list of floats over the ranks (every rank gets the result)."""
        rows = self.allgather([float(v) for v in values])
        fn = {"max": max, "min": min, "sum": sum}[op]
        return [float(fn(col)) for col in zip(*rows)]

    def close(self):
        for s in list(self.peers.values()) + ([self.hub] if self.hub else []):
            try:
                s.close()
            except OSError:
                pass
        self.peers, self.hub = {}, None
