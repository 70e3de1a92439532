"""Sample-axis sharding of the greedy loop: one process per GPU, each holding a contiguous
block of sample columns and a full replica of the covered mask.

One iteration = local scoring -> every rank's best (score, global idx, new_count) record reaches every rank ->
the same winner everywhere (score descending, index ascending = np.argmax's first maximum,
utmos/select.py:48) -> the winner's column reaches every rank -> covered |= column.

`connect_shards` picks how the records and the column travel, collectively and in this order:

  mailboxes  every rank maps every other rank's columns and a small record mailbox (hipIpc); after a self-test the
             loop runs on the devices alone -- 64-byte records stored straight into the peers' mailboxes, the winner's
             column read from a one-time local copy of the peers' columns (when HBM allows) or in place over xGMI;
  rccl       one ncclAllGather of the records and one ncclBroadcast of the winner's column from its owner per
             iteration (north_star's protocol; what `--exchange rccl` forces); `rccl-allreduce` is its root-free
             variant and the fallback `auto` takes: the column travels by an ncclAllReduce(sum) of
             owner's-column-else-zeros, so the host never has to learn the root (no stream sync per iteration);
  otherwise  an error: there is no host-staged product path.

`sharded_greedy` drives the same protocol from the host through the C ABI's building blocks (utm_local_best /
utm_get_column / utm_apply_records) over any object with `allgather` and `broadcast`; the CPU tests run it over
torch.distributed (gloo) with stand-in shards, and it documents what the device-side forms compute.
"""
import atexit
import os
import socket
import struct
import time

import numpy as np


def shard_bounds(n_samples, rank, world):
    """Contiguous, near-equal blocks; lower rank = lower sample indices (keeps the tie-break global)."""
    first = rank * n_samples // world
    return first, (rank + 1) * n_samples // world - first


def pick_winner(records):
    """records: [(score, idx, new)] in rank order -> winning rank or None."""
    best = None
    for rank, (score, idx, _new) in enumerate(records):
        if idx < 0:
            continue
        if best is None or score > records[best][0] or (score == records[best][0] and idx < records[best][1]):
            best = rank
    return best


class TorchDistTransport:
    """allgather/broadcast over torch.distributed (gloo on CPU tensors)."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def allgather(self, record):
        t = self.torch.tensor([float(record[0]), float(record[1]), float(record[2])], dtype=self.torch.float64)
        out = [self.torch.zeros(3, dtype=self.torch.float64) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return [(float(o[0]), int(o[1]), int(o[2])) for o in out]

    def broadcast(self, column, n_words, src):
        buf = self.torch.from_numpy(column.view(np.int64)) if column is not None \
            else self.torch.zeros(n_words, dtype=self.torch.int64)
        self.dist.broadcast(buf, src=src, group=self.group)
        return buf.numpy().view(np.uint64)


class SocketTransport:
    """allgather/broadcast over plain TCP sockets (star through rank 0): the start-up channel of a launch (handles,
    ids, agreement on the exchange, end-of-run barrier).  No third-party dependency, so GPU processes never load a
    second HIP runtime next to libutmos_hip.so.  `hello` is the launch's nonce: rank 0 only admits peers that
    present it, a peer only stays with a rank 0 that acknowledges it."""

    def __init__(self, rank, world, addr="127.0.0.1", port=29617, timeout=300.0, listener=None, hello=b""):
        self.rank, self.world = rank, world
        self.peers = []
        if world == 1:
            return
        if rank == 0:
            srv = listener            # rank 0 may hand in a socket it already bound (e.g. to an ephemeral port)
            if srv is None:
                srv = socket.socket()
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout)
            by_rank = {}
            while len(by_rank) < world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                try:
                    conn.settimeout(timeout)
                    peer = struct.unpack("<i", self._recv(conn, 4))[0]
                    if self._recv(conn, len(hello)) != hello or not 0 < peer < world or peer in by_rank:
                        raise ConnectionError("not a peer of this launch")
                    conn.sendall(b"\x01")
                    conn.settimeout(None)
                    by_rank[peer] = conn
                except (OSError, ConnectionError, struct.error):
                    conn.close()
            srv.close()
            self.peers = [by_rank[r] for r in range(1, world)]
        else:
            deadline = time.time() + timeout
            conn = self.try_connect(rank, addr, port, hello, timeout)
            while conn is None and time.time() < deadline:      # rank 0 may not be listening yet
                time.sleep(0.05)
                conn = self.try_connect(rank, addr, port, hello, timeout)
            if conn is None:
                raise ConnectionError(f"rank {rank}: no rank 0 of this launch at {addr}:{port}")
            self.peers = [conn]

    @staticmethod
    def try_connect(rank, addr, port, hello, timeout):
        """One attempt to join rank 0 at addr:port; None when nobody answers there or the nonce is not accepted."""
        try:
            conn = socket.create_connection((addr, port), timeout=min(timeout, 10.0))
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            conn.sendall(struct.pack("<i", rank) + hello)
            conn.settimeout(timeout)
            if conn.recv(1) != b"\x01":
                conn.close()
                return None
            conn.settimeout(None)
            return conn
        except OSError:
            return None

    @staticmethod
    def _recv(conn, n):
        buf = bytearray()
        while len(buf) < n:
            part = conn.recv(n - len(buf))
            if not part:
                raise ConnectionError("peer closed the connection")
            buf += part
        return bytes(buf)

    def allgather(self, record):
        pack = struct.Struct("<dqq")
        mine = pack.pack(float(record[0]), int(record[1]), int(record[2]))
        return [pack.unpack(b) for b in self.allgather_bytes(mine)]

    def allgather_bytes(self, blob):
        """Equal-sized byte strings from every rank, in rank order."""
        if self.world == 1:
            return [blob]
        n = len(blob)
        if self.rank == 0:
            parts = [blob] + [self._recv(c, n) for c in self.peers]
            for c in self.peers:
                c.sendall(b"".join(parts))
            return parts
        self.peers[0].sendall(blob)
        data = self._recv(self.peers[0], n * self.world)
        return [data[i * n:(i + 1) * n] for i in range(self.world)]

    def agree(self, ok):
        """True on every rank iff `ok` is true on every rank (collective)."""
        return all(r[1] == 1 for r in self.allgather((0.0, 1 if ok else 0, 0)))

    def barrier(self):
        self.allgather((0.0, 0, 0))

    def broadcast(self, column, n_words, src):
        n = n_words * 8
        if self.rank == 0:
            data = column.tobytes() if src == 0 else self._recv(self.peers[src - 1], n)
            for r, c in enumerate(self.peers, start=1):
                if r != src:
                    c.sendall(data)
        else:
            if self.rank == src:
                self.peers[0].sendall(column.tobytes())
                return column
            data = self._recv(self.peers[0], n)
        return np.frombuffer(data, dtype=np.uint64).copy()

    def close(self):
        for c in self.peers:
            c.close()
        self.peers = []


def dist_env():
    """(rank, world, local_rank) of a one-process-per-GPU launch (torchrun-style environment), else (0, 1, 0)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))))


# ----------------------------------------------------------------------------- start-up rendezvous
NONCE_BYTES = 16
_BOOT = struct.Struct(f"<i{NONCE_BYTES}s128s")      # port, nonce, ncclUniqueId


def rendezvous_file():
    """Where rank 0 publishes {port, nonce, ncclUniqueId}.  UTMOS_RENDEZVOUS_FILE names it outright (launchers whose
    ranks do not share a parent process); otherwise MASTER_PORT + run id + the launcher's pid."""
    explicit = os.environ.get("UTMOS_RENDEZVOUS_FILE")
    if explicit:
        return explicit
    key = f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"utmos_amd_rendezvous_{key}")


def _publish(path, payload):
    tmp = f"{path}.{os.getpid()}"
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
    with os.fdopen(fd, "wb") as fh:
        fh.write(payload)
    os.replace(tmp, path)


def _remove(path):
    try:
        os.remove(path)
    except OSError:
        pass


def bootstrap(rank, world, make_id=None, timeout=300.0):
    """One-node start-up: rank 0 binds an ephemeral TCP port and publishes {port, nonce, ncclUniqueId}; the other
    ranks read the file, connect and present the nonce.  A file left by a crashed launch names a dead port or a
    nonce nobody acknowledges: the reader then re-reads the file instead of insisting.  -> (transport, uid)."""
    if world == 1:
        return SocketTransport(0, 1), (make_id() if make_id else None)
    path = rendezvous_file()
    if rank == 0:
        srv = socket.socket()
        srv.bind(("127.0.0.1", 0))
        uid = make_id() if make_id else bytes(128)
        nonce = os.urandom(NONCE_BYTES)
        _publish(path, _BOOT.pack(srv.getsockname()[1], nonce, uid))
        atexit.register(_remove, path)
        try:
            transport = SocketTransport(0, world, listener=srv, timeout=timeout, hello=nonce)
        finally:
            _remove(path)
        return transport, (uid if make_id else None)
    started = time.time()
    while time.time() - started < timeout:
        try:
            if os.path.getmtime(path) >= started - 600:          # never a leftover of a much older launch
                with open(path, "rb") as fh:
                    data = fh.read()
                if len(data) == _BOOT.size:
                    port, nonce, uid = _BOOT.unpack(data)
                    conn = SocketTransport.try_connect(rank, "127.0.0.1", port, nonce, timeout)
                    if conn is not None:
                        transport = SocketTransport(rank, 1)
                        transport.world, transport.peers = world, [conn]
                        return transport, (uid if make_id else None)
        except OSError:
            pass
        time.sleep(0.05)
    raise RuntimeError(f"rank {rank}: no rank 0 found through {path} within {timeout:.0f} s")


# ----------------------------------------------------------------------------- choosing the exchange
def enable_mailboxes(shard, transport):
    """hipIpc mappings of every shard's columns + record mailboxes, self-tested; every step agreed on by all ranks.
    Collective.  True when run() now exchanges through the mailboxes."""
    if transport.world == 1:
        return False
    try:
        blob = shard.p2p_export()
    except Exception:  # noqa: BLE001 - any failure means "no hipIpc here"
        blob = None
    size = max(r[1] for r in transport.allgather((0.0, -1 if blob is None else len(blob), 0)))
    blobs = transport.allgather_bytes(blob if blob is not None else bytes(max(size, 1)))
    ok = blob is not None
    if ok:
        try:
            shard.p2p_import(transport.rank, blobs)
        except Exception:  # noqa: BLE001
            ok = False
    if not transport.agree(ok):
        shard.p2p = False          # (a context that imported keeps its mappings; nothing uses them)
        return False
    try:
        boxes = shard.p2p_selftest()
    except Exception:  # noqa: BLE001
        boxes = False
    if not transport.agree(boxes):
        return False
    shard.p2p_use_mailboxes(True)
    return True


def connect_shards(shard, transport, uid, exchange="auto"):
    """Make shard.run() collective.  exchange: 'auto' (mailboxes, else RCCL), 'mailboxes', 'rccl', 'rccl-allreduce'.
    Returns the name of the exchange in effect; raises RuntimeError when none can be set up on every rank."""
    if transport.world == 1:
        return "none"
    if exchange in ("auto", "mailboxes") and enable_mailboxes(shard, transport):
        return "mailboxes"
    if exchange == "mailboxes":
        raise RuntimeError("the record mailboxes could not be set up on every shard")
    err = None
    try:
        shard.comm_init(transport.rank, transport.world, uid)
    except Exception as exc:  # noqa: BLE001
        err = exc
    if not transport.agree(err is None):
        raise RuntimeError(f"no exchange between the shards: hipIpc mailboxes unavailable and RCCL did not come up ({err})")
    if exchange in ("rccl-allreduce", "auto"):
        # the fallback of `auto` is the root-free form: a broadcast's root is only known on the host after a stream sync
        # per iteration (measured with one rank at 10M x 2,504: +33 us per iteration, against +2.6 us for this form)
        shard.comm_column_by_allreduce(True)
        return "rccl-allreduce"
    return "rccl"


def sharded_greedy(shard, transport, select_count):
    """The protocol driven from the host, one building-block call at a time (tests, documentation).

    Yields (global idx, new_count, score) per selected sample; identical on every rank.  `shard` is this rank's
    matrix (first_sample/n_local set, state and weights applied, reset done)."""
    n_words = shard.column_words()
    p2p = getattr(shard, "p2p", False)   # remote winners' columns are read in place: nothing to broadcast
    for _ in range(int(select_count)):
        records = transport.allgather(shard.local_best())
        owner = pick_winner(records)
        if owner is None:
            return
        column = None
        if transport.world > 1 and not p2p:
            mine = shard.get_column(records[owner][1]) if transport.rank == owner else None
            column = transport.broadcast(mine, n_words, owner)
        out = shard.apply_records(records, None if transport.rank == owner else column)
        if out is None:
            return
        yield out
