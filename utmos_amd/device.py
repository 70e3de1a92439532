"""HBM-resident bit-packed presence matrix + the device-resident greedy loop (thin object layer
over the C ABI).  One DeviceMatrix = one GPU = one contiguous shard of the sample axis."""
import ctypes

import numpy as np

from . import _native as nat


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class DeviceMatrix:
    """variants x samples presence matrix, column-major bitsets in HBM, split into chunks along
    the variant axis (the role the reference's row-chunked hdf5 plays, utmos/select.py:198-231)."""

    def __init__(self, n_samples, device=0, first_sample=0, n_local=None, profile_events=False,
                 af_sequential=False):
        self._h = ctypes.c_void_p()
        self.n_samples = int(n_samples)
        self.first_sample = int(first_sample)
        self.n_local = int(self.n_samples - self.first_sample if n_local is None else n_local)
        self.chunk_vars = []
        self.p2p = False
        self.fused = False      # run() is collective over the shards (mailboxes or RCCL)
        self.fused_mailboxes = False
        flags = (nat.FLAG_PROFILE_EVENTS if profile_events else 0) | (nat.FLAG_AF_SEQUENTIAL if af_sequential else 0)
        code = nat.lib().utm_ctx_create(int(device), self.n_samples, self.first_sample, self.n_local, flags,
                                        ctypes.byref(self._h))
        if code != nat.UTM_OK:
            msg = nat.lib().utm_last_error().decode()
            if self._h:
                nat.lib().utm_ctx_destroy(self._h)
                self._h = ctypes.c_void_p()
            raise nat.NativeError(code, msg)

    # -- lifetime
    def close(self):
        if self._h:
            nat.lib().utm_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover - interpreter shutdown
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- shape, as the reference reads it (matrix.shape[0] / [1], select.py:89, :155)
    @property
    def shape(self):
        return (int(sum(self.chunk_vars)), self.n_samples)

    # -- matrix
    def add_chunk(self, n_var):
        idx = ctypes.c_int32(-1)
        nat.check(nat.lib().utm_add_chunk(self._h, int(n_var), ctypes.byref(idx)))
        self.chunk_vars.append(int(n_var))
        return idx.value

    def upload_columns(self, chunk, cols, first_col=0):
        cols = np.ascontiguousarray(cols, dtype=np.uint64)
        nat.check(nat.lib().utm_upload_columns(self._h, chunk, first_col, cols.shape[0], _ptr(cols), cols.shape[1]))

    def upload_rows_packed(self, chunk, rows, first_var=0):
        """rows: uint8 (n, >= ceil(S/8)), numpy.packbits(axis=1) order (utmos/convert.py:85)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        nat.check(nat.lib().utm_upload_rows_packed(self._h, chunk, int(first_var), rows.shape[0], _ptr(rows), rows.shape[1]))

    def download_columns(self, chunk, first_col=0, n_cols=None, out=None):
        """Columns [first_col, first_col + n_cols) of a chunk as uint64 (n_cols, ceil(n_var/64)); `out` = a C-contiguous
        array (or slice of whole rows of one) to fill instead of a new one -- large matrices come down in slices."""
        n_cols = self.n_local - first_col if n_cols is None else n_cols
        words = (self.chunk_vars[chunk] + 63) // 64
        if out is None:
            out = np.empty((n_cols, words), dtype=np.uint64)
        elif out.dtype != np.uint64 or out.shape != (n_cols, words) or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous uint64 array of shape (n_cols, words)")
        nat.check(nat.lib().utm_download_columns(self._h, chunk, first_col, n_cols, _ptr(out), words))
        return out

    def var_count(self):
        out = np.zeros(self.n_local, dtype=np.int64)
        nat.check(nat.lib().utm_var_count(self._h, _ptr(out)))
        return out

    def synth_fill(self, chunk, seed=0, first_var_global=0):
        nat.check(nat.lib().utm_synth_fill(self._h, chunk, int(seed), int(first_var_global)))

    # -- options
    def set_state(self, state):
        st = np.ascontiguousarray(state, dtype=np.uint8)
        if st.shape != (self.n_samples,):
            raise ValueError(f"state must have {self.n_samples} entries")
        nat.check(nat.lib().utm_set_sample_state(self._h, _ptr(st)))

    def set_weights(self, weights):
        if weights is None:
            nat.check(nat.lib().utm_set_weights(self._h, None))
            return
        w = np.ascontiguousarray(weights, dtype=np.float64)
        if w.shape != (self.n_samples,):
            raise ValueError(f"weights must have {self.n_samples} entries")
        nat.check(nat.lib().utm_set_weights(self._h, _ptr(w)))

    def set_af(self, chunk, af):
        """af: float32 (hdf5 semantics) or float64 (in-memory semantics) per variant; None clears all."""
        if af is None:
            nat.check(nat.lib().utm_set_af(self._h, 0, nat.AF_NONE, None))
            return
        af = np.ascontiguousarray(af).reshape(-1)
        if af.dtype == np.float32:
            mode = nat.AF_F32
        elif af.dtype == np.float64:
            mode = nat.AF_F64
        else:
            raise TypeError("AF must be float32 or float64")
        if af.shape[0] != self.chunk_vars[chunk]:
            raise ValueError("AF length differs from the chunk's variant count")
        nat.check(nat.lib().utm_set_af(self._h, chunk, mode, _ptr(af)))

    # -- loop
    def reset(self):
        nat.check(nat.lib().utm_reset(self._h))

    def step(self):
        """One greedy iteration.  (idx, new_count, score) or None for the reference's (None, None)."""
        i, n, s = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double()
        nat.check(nat.lib().utm_step(self._h, ctypes.byref(i), ctypes.byref(n), ctypes.byref(s)))
        return None if i.value < 0 else (i.value, n.value, s.value)

    def run(self, k_max):
        k_max = int(k_max)
        idx = np.zeros(max(k_max, 1), dtype=np.int64)
        new = np.zeros(max(k_max, 1), dtype=np.int64)
        score = np.zeros(max(k_max, 1), dtype=np.float64)
        done = ctypes.c_int64(0)
        nat.check(nat.lib().utm_run(self._h, k_max, _ptr(idx), _ptr(new), _ptr(score), ctypes.byref(done)))
        n = done.value
        return idx[:n].copy(), new[:n].copy(), score[:n].copy()

    def peek_scores(self):
        counts = np.zeros(self.n_local, dtype=np.int64)
        scores = np.zeros(self.n_local, dtype=np.float64)
        nat.check(nat.lib().utm_peek_scores(self._h, _ptr(counts), _ptr(scores)))
        return counts, scores

    def peek_estimates(self):
        """AF in fixed point: counts and scores of the next iteration from the parallel full pass (not the chains)."""
        counts = np.zeros(self.n_local, dtype=np.int64)
        scores = np.zeros(self.n_local, dtype=np.float64)
        nat.check(nat.lib().utm_peek_estimates(self._h, _ptr(counts), _ptr(scores)))
        return counts, scores

    def covered(self, chunk):
        out = np.zeros((self.chunk_vars[chunk] + 63) // 64, dtype=np.uint64)
        nat.check(nat.lib().utm_get_covered(self._h, chunk, _ptr(out)))
        return out

    def stats(self):
        st = nat.Stats()
        nat.check(nat.lib().utm_get_stats(self._h, ctypes.byref(st)))
        return {name: getattr(st, name) for name, _ in nat.Stats._fields_}

    def set_decremental(self, on, threshold=0.0):
        """Allow decremental scoring for later iterations (exact; fewer bytes; reported separately)."""
        nat.check(nat.lib().utm_set_decremental(self._h, 1 if on else 0, float(threshold)))

    def set_af_exact_scores(self, on):
        """AF modes: off = skip the sequential chain of an unambiguous winner (same rows; scores become estimates)."""
        nat.check(nat.lib().utm_set_af_exact_scores(self._h, 1 if on else 0))

    def stream_calibration(self, launches=20):
        """GB/s of plain streaming reads over the resident columns (the scoring kernel's access shape, nothing else)."""
        out = ctypes.c_double(0.0)
        nat.check(nat.lib().utm_stream_calibration(self._h, int(launches), ctypes.byref(out)))
        return out.value

    def set_profile(self, on):
        nat.check(nat.lib().utm_set_profile(self._h, 1 if on else 0))

    # -- sharded building blocks
    def local_best(self):
        rec = nat.Record()
        nat.check(nat.lib().utm_local_best(self._h, ctypes.byref(rec)))
        return (rec.score, rec.idx, rec.new_count)

    def column_words(self):
        n = ctypes.c_uint64()
        nat.check(nat.lib().utm_column_words(self._h, ctypes.byref(n)))
        return n.value

    def get_column(self, global_idx):
        out = np.zeros(self.column_words(), dtype=np.uint64)
        nat.check(nat.lib().utm_get_column(self._h, int(global_idx), _ptr(out)))
        return out

    def apply_records(self, records, winner_col=None):
        """records: [(score, idx, new_count)] of every shard in rank order."""
        arr = (nat.Record * len(records))()
        for r, (score, idx, new) in zip(arr, records):
            r.score, r.idx, r.new_count = float(score), int(idx), int(new)
        col = None if winner_col is None else np.ascontiguousarray(winner_col, dtype=np.uint64)
        i, n, s = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double()
        nat.check(nat.lib().utm_apply_records(self._h, arr, len(records), _ptr(col), ctypes.byref(i),
                                              ctypes.byref(n), ctypes.byref(s)))
        return None if i.value < 0 else (i.value, n.value, s.value)

    # -- P2P column access between shards (hipIpc)
    def p2p_export(self):
        n = ctypes.c_uint64()
        nat.check(nat.lib().utm_p2p_blob_bytes(self._h, ctypes.byref(n)))
        buf = ctypes.create_string_buffer(n.value)
        nat.check(nat.lib().utm_p2p_export(self._h, buf))
        return buf.raw

    def p2p_import(self, rank, blobs):
        """blobs: every shard's p2p_export() in rank order.  Remote winners are then read in place."""
        joined = b"".join(blobs)
        buf = ctypes.create_string_buffer(joined, len(joined))
        nat.check(nat.lib().utm_p2p_import(self._h, int(rank), len(blobs), buf))
        self.p2p = True

    def p2p_selftest(self):
        """Collective over the shards: can this shard see every peer's mailbox stores?"""
        ok = ctypes.c_int32(0)
        nat.check(nat.lib().utm_p2p_selftest(self._h, ctypes.byref(ok)))
        return bool(ok.value)

    def p2p_use_mailboxes(self, on=True):
        """After every shard's self-test passed: run() becomes collective, records travel through the mailboxes."""
        nat.check(nat.lib().utm_p2p_use_mailboxes(self._h, 2 if on == "single" else 1 if on else 0))
        self.fused_mailboxes = bool(on)
        self.fused = bool(on) or getattr(self, "has_comm", False)

    def exchange(self):
        """How run() meets the other shards right now: 'none', 'mailboxes', 'rccl' or 'caller-driven'."""
        return nat.EXCHANGE_NAMES[self.stats()["exchange"]]

    # -- RCCL
    @staticmethod
    def comm_unique_id():
        buf = ctypes.create_string_buffer(nat.UNIQUE_ID_BYTES)
        nat.check(nat.lib().utm_comm_get_unique_id(buf))
        return buf.raw

    def comm_init(self, rank, n_ranks, unique_id):
        """One RCCL communicator over the shards.  Unless the mailboxes are switched on, run() then exchanges through
        it: ncclAllGather of the records, ncclBroadcast of the winner's column from its owner."""
        buf = ctypes.create_string_buffer(bytes(unique_id), nat.UNIQUE_ID_BYTES)
        # RCCL announces itself ("RCCL version : ...") on stdout when its first communicator comes up; stdout belongs to
        # the caller (the TSV of `select`, bench.py's one JSON line): that banner goes to stderr instead
        import os
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            code = nat.lib().utm_comm_init(self._h, int(rank), int(n_ranks), buf)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        nat.check(code)
        self.fused = True
        self.has_comm = True

    def comm_column_by_allreduce(self, on=True):
        """RCCL exchange: winner column by a root-free ncclAllReduce(sum) instead of an ncclBroadcast from its owner
        (no host sync per iteration; about twice the bytes on the links).  Same rows."""
        nat.check(nat.lib().utm_comm_column_by_allreduce(self._h, 1 if on else 0))

    def allreduce_max(self, value):
        v = ctypes.c_double(float(value))
        nat.check(nat.lib().utm_comm_allreduce_max(self._h, ctypes.byref(v)))
        return v.value


def synth_host(seed, n_var, n_samples, first_sample=0, n_samp=None, first_var_global=0, want_cols=True, want_af=True):
    """The device generator's host twin (bit-identical; no GPU call)."""
    n_samp = n_samples - first_sample if n_samp is None else n_samp
    words = (n_var + 63) // 64
    cols = np.zeros((n_samp, words), dtype=np.uint64) if want_cols else None
    af = np.zeros(n_var, dtype=np.float32) if want_af else None
    nat.check(nat.lib().utm_synth_host(int(seed), int(first_var_global), int(n_var), int(n_samples), int(first_sample),
                                       int(n_samp), _ptr(cols), words, _ptr(af)))
    return cols, af
