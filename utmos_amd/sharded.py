"""Sample-axis sharding of the greedy loop.

Each rank holds a contiguous block of sample columns and a full replica of the covered mask.  One
iteration = local scoring -> every rank's best (score, global idx, new_count) record exchanged ->
the same winner chosen on every rank (score descending, index ascending = np.argmax's first maximum,
utmos/select.py:48) -> the winner's column reaches every rank -> covered |= column.

Two transports:
  * fused (production): `DeviceMatrix.comm_init` + `run()` -- one ncclAllGather per iteration carrying
    {record, candidate column} of every rank, all on the GPU stream, no host round trip;
  * host staged (this module): the same protocol driven from Python through the C ABI's building
    blocks (utm_local_best / utm_get_column / utm_apply_records) over any object with `allgather`
    and `broadcast` -- used with torch.distributed (gloo) in tests and on hosts without RCCL.
"""
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
    """allgather/broadcast over plain TCP sockets (star through rank 0); no third-party dependency,
    so GPU processes never have to load a second HIP runtime next to libutmos_hip.so."""

    def __init__(self, rank, world, addr="127.0.0.1", port=29617, timeout=300.0, listener=None):
        import socket
        import struct
        import time
        self.rank, self.world, self._struct = rank, world, struct
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
                by_rank[struct.unpack("<i", self._recv(conn, 4))[0]] = conn
            srv.close()
            self.peers = [by_rank[r] for r in range(1, world)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    conn = socket.create_connection((addr, port), timeout=timeout)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            conn.sendall(struct.pack("<i", rank))
            self.peers = [conn]

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
        pack = self._struct.Struct("<dqq")
        mine = pack.pack(float(record[0]), int(record[1]), int(record[2]))
        if self.world == 1:
            return [pack.unpack(mine)]
        if self.rank == 0:
            blob = mine + b"".join(self._recv(c, pack.size) for c in self.peers)
            for c in self.peers:
                c.sendall(blob)
        else:
            self.peers[0].sendall(mine)
            blob = self._recv(self.peers[0], pack.size * self.world)
        return [pack.unpack_from(blob, r * pack.size) for r in range(self.world)]

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


def dist_env():
    """(rank, world, local_rank) of a one-process-per-GPU launch (torchrun-style environment), else (0, 1, 0)."""
    import os
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))))


def _rendezvous_path():
    import os
    key = f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"utmos_amd_rendezvous_{key}")


def _publish(path, payload):
    import os
    tmp = f"{path}.{os.getpid()}"
    with open(tmp, "wb") as fh:
        fh.write(payload)
    os.replace(tmp, path)


def _await(path, n_bytes, rank, timeout):
    import os
    import time
    started = time.time()
    while time.time() - started < timeout:
        try:
            if os.path.getmtime(path) >= started - 600:      # never a leftover of an older launch
                with open(path, "rb") as fh:
                    data = fh.read()
                if len(data) == n_bytes:
                    return data
        except FileNotFoundError:
            pass
        time.sleep(0.05)
    raise RuntimeError(f"rank {rank}: nothing at {path} after {timeout:.0f} s")


def rendezvous_unique_id(rank, make_id, timeout=300.0):
    """Share rank 0's 128-byte ncclUniqueId through a file every rank of this launch can name
    (same MASTER_PORT, run id and parent process = the launcher)."""
    path = _rendezvous_path() + "_id"
    if rank == 0:
        uid = make_id()
        _publish(path, uid)
        return uid, path
    return _await(path, 128, rank, timeout), path


def bootstrap(rank, world, make_id=None, timeout=300.0):
    """One-node start-up of a one-process-per-GPU launch: rank 0 binds a TCP socket to an ephemeral port and
    publishes {port, ncclUniqueId} in the launch's rendezvous file; everyone connects.  -> (transport, uid)."""
    import socket
    import struct
    if world == 1:
        return SocketTransport(0, 1), (make_id() if make_id else None)
    path = _rendezvous_path() + "_boot"
    if rank == 0:
        srv = socket.socket()
        srv.bind(("127.0.0.1", 0))
        uid = make_id() if make_id else bytes(128)
        _publish(path, struct.pack("<i", srv.getsockname()[1]) + uid)
        transport = SocketTransport(0, world, listener=srv, timeout=timeout)
        try:
            import os
            os.remove(path)
        except OSError:
            pass
        return transport, (uid if make_id else None)
    data = _await(path, 132, rank, timeout)
    port = struct.unpack("<i", data[:4])[0]
    return SocketTransport(rank, world, "127.0.0.1", port, timeout=timeout), (data[4:] if make_id else None)


def enable_p2p(shard, transport):
    """Give every shard access to every other shard's columns and a device-side record exchange.  Collective.
    In order of preference: hipIpc mappings of the columns (+ a one-time local copy when HBM allows) with mailboxes
    in uncached device memory; mailboxes in host shared memory instead; and, where device memory cannot be shared
    between processes at all, a local copy of the peers' columns filled through a host shared-memory file.
    Returns True when winner columns need no broadcast any more; shard.fused tells whether run() became collective."""
    import os
    if transport.world == 1:
        return False
    mapped = False
    if os.environ.get("UTM_NO_IPC", "0") == "0":
        try:
            blob = shard.p2p_export()
        except Exception:  # noqa: BLE001 - any failure means "no hipIpc here"
            blob = None
        size = max(transport.allgather((0.0, -1 if blob is None else len(blob), 0)), key=lambda r: r[1])[1]
        blobs = transport.allgather_bytes(blob if blob is not None else bytes(max(size, 1)))
        ok = blob is not None
        if ok:
            try:
                shard.p2p_import(transport.rank, blobs)
            except Exception:  # noqa: BLE001
                ok = False
        mapped = all(r[1] == 1 for r in transport.allgather((0.0, 1 if ok else 0, 0)))
    if not mapped:
        shard.p2p = False   # (a context that imported keeps its mappings but they are not used)
        if not _replicate_via_host(shard, transport):
            return False
    # every shard can resolve every winner's column: can the shards also exchange their records on the device?
    boxes = shard.p2p_selftest() if mapped and os.environ.get("UTM_MBOX", "device") != "host" else False
    if all(r[1] == 1 for r in transport.allgather((0.0, 1 if boxes else 0, 0))):
        shard.p2p_use_mailboxes(True)   # shard.fused: run() is now collective, nothing goes through the host
    elif _host_mailboxes(shard, transport):
        shard.p2p_use_mailboxes(True)   # same loop, the 64-byte records cross PCIe instead of xGMI
    return True


def _shared_file(transport, size):
    """A zero-filled shared-memory file of `size` bytes that every rank has mapped.  Collective; -> mmap or None.
    (Unlinked as soon as everybody has it open: the mappings keep it alive.)"""
    import mmap
    import os
    size = (size + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE
    name = b""
    if transport.rank == 0:
        path = f"/dev/shm/utmos_amd_{os.getpid()}_{int.from_bytes(os.urandom(4), 'little')}"
        try:
            room = os.statvfs("/dev/shm")
            if room.f_bavail * room.f_frsize > size + (1 << 30):
                with open(path, "wb") as fh:
                    fh.truncate(size)
                name = path.encode()
        except OSError:
            name = b""
    name = transport.allgather_bytes(name.ljust(96, b"\0"))[0].rstrip(b"\0").decode()
    shared = None
    if name:
        try:
            with open(name, "r+b") as fh:
                shared = mmap.mmap(fh.fileno(), size)
        except (OSError, ValueError):
            shared = None
    everyone = all(r[1] == 1 for r in transport.allgather((0.0, 1 if shared is not None else 0, 0)))
    if transport.rank == 0 and name:
        try:
            os.unlink(name)
        except OSError:
            pass
    return shared if everyone else None


def _replicate_via_host(shard, transport):
    """No hipIpc: every shard writes its columns into a host shared-memory file and uploads the others' from there
    into a local copy (one-time cost ~ the matrix over PCIe; needs the matrix to fit into /dev/shm and a second time
    into every GPU).  Collective; True when every shard holds its copy."""
    import numpy as np
    ranges = transport.allgather((0.0, shard.first_sample, shard.n_local))
    firsts, locals_ = [r[1] for r in ranges], [r[2] for r in ranges]
    words = [(int(v) + 63) // 64 for v in shard.chunk_vars]
    offsets = np.concatenate([[0], np.cumsum([w * 8 * shard.n_samples for w in words])]).astype(np.int64)
    shared = _shared_file(transport, int(offsets[-1]))
    if shared is None:
        return False
    views = [np.frombuffer(shared, dtype=np.uint64, count=shard.n_samples * w, offset=int(off)).reshape(shard.n_samples, w)
             for w, off in zip(words, offsets)]
    ok = True
    try:
        for k in range(len(views)):
            views[k][shard.first_sample:shard.first_sample + shard.n_local] = shard.download_columns(k)
    except Exception:  # noqa: BLE001
        ok = False
    ok = all(r[1] == 1 for r in transport.allgather((0.0, 1 if ok else 0, 0)))      # also: everybody's columns are in
    if ok:
        try:
            shard.p2p_replica_from_host(transport.rank, firsts, locals_, views)
        except Exception:  # noqa: BLE001 - e.g. no room for the copy
            ok = False
    ok = all(r[1] == 1 for r in transport.allgather((0.0, 1 if ok else 0, 0)))      # also: nobody reads the file any more
    del views
    try:
        shared.close()
    except BufferError:
        pass
    return ok


def _host_mailboxes(shard, transport):
    """Fallback for the record exchange: mailboxes in a POSIX shared-memory file mapped by every shard's process and
    page-locked for its GPU.  Collective; True when every shard passed the self-test through them."""
    shared = _shared_file(transport, shard.p2p_host_mailbox_bytes(transport.world))
    ok = shared is not None
    if ok:
        try:
            shard.p2p_host_mailboxes(shared)
        except Exception:  # noqa: BLE001 - any failure means "not here"
            ok = False
    if not all(r[1] == 1 for r in transport.allgather((0.0, 1 if ok else 0, 0))):
        return False
    boxes = shard.p2p_selftest()
    return all(r[1] == 1 for r in transport.allgather((0.0, 1 if boxes else 0, 0)))


def sharded_greedy(shard, transport, select_count):
    """Yield (global idx, new_count, score) per selected sample; identical on every rank.

    `shard` is this rank's matrix (DeviceMatrix with first_sample/n_local set, state and weights
    already applied, reset done)."""
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
