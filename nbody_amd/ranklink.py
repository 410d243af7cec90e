"""ranklink -- rendezvous of the ranks of ONE node over a Unix-domain socket hub (stdlib only).

bench.py under `torch.distributed.run` is one Python process per GPU.  Importing torch there would load the HIP runtime
and the librccl bundled with the torch wheel before libnbody_hip.so binds its own (/opt/rocm's), so the few host-side
collectives the harness needs -- barrier, max/min/sum of some floats, a broadcast of RCCL's 128-byte unique id, an
all-gather of byte rows for the host transport -- go through this instead: rank 0 listens on an abstract socket named
after the launcher's MASTER_PORT and run id, the other ranks connect, and every collective is "send mine to rank 0,
get everybody's back".  The data path (the per-step all-gather of source positions) is RCCL inside the library and never
passes through here.  The C harness (nbody-bench --gpus P) does the same over a shared page (csrc/rank_page.c).
"""
import os
import pickle
import socket
import struct
import time


class RankLink:
    def __init__(self, rank, world, name=None, timeout_s=900.0):
        self.rank, self.world, self.timeout_s = rank, world, timeout_s
        self.peers, self.hub = {}, None
        if world == 1:
            return
        if name is None:
            name = "nbody_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"))
        addr = "\0" + name   # abstract namespace: nothing to unlink, gone with the last descriptor
        deadline = time.monotonic() + timeout_s
        if rank == 0:
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(addr)
            srv.listen(world)
            srv.settimeout(timeout_s)
            while len(self.peers) < world - 1:
                conn, _ = srv.accept()
                conn.settimeout(timeout_s)
                self.peers[self._recv(conn)] = conn
            srv.close()
            assert sorted(self.peers) == list(range(1, world)), sorted(self.peers)
        else:
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(addr)
                    break
                except (ConnectionRefusedError, FileNotFoundError):
                    s.close()
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"rank {rank}: no hub at {name!r} within {timeout_s} s")
                    time.sleep(0.02)
            s.settimeout(timeout_s)
            self.hub = s
            self._send(s, rank)

    @staticmethod
    def _send(sock, obj):
        data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
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
        return pickle.loads(exactly(n))

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
