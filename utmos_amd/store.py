"""`.utm` packed matrix store -- the on-disk form of the HBM-resident matrix; takes the place of the reference's
row-chunked hdf5 (`write_append_hdf5` / `add_varcount_to_h5`, utmos/select.py:198-238; reuse :250-251, :406-415).

Layout (little endian, every block at a 4 KiB boundary so that a block maps straight into host memory):

    0       header, 4096 bytes: magic "UTMSTORE", u32 version, u32 n_samples, u32 n_chunks, u32 has_af,
            u64 offset/length of the sample names, u64 offset of var_count, u64 offset of the chunk table
    ...     per chunk, in the order the chunks were flushed while loading:
              AF, float32[n_var]   (only with --af; float32 is what the reference's hdf5 holds, select.py:218-223)
              columns, uint64[n_samples][W], W = ceil(n_var / 64): sample s, variant v = bit (v & 63) of word (v >> 6)
    ...     sample names (UTF-8, newline separated), var_count int64[n_samples],
            chunk table: per chunk 4 x u64 {n_var, W, AF offset (0 = none), column offset}

A writer appends one chunk at a time, as `load_files` flushes it, and never holds more than that chunk on the host;
names, var_count and the table follow once the last chunk is in, and the header is written last (a file without a
valid header is an interrupted write).  A reader maps (numpy.memmap) exactly the column range it owns: a shard of a
multi-GPU run reads n_local / n_samples of every chunk, nothing else.
"""
import struct

import numpy as np

MAGIC = b"UTMSTORE"
VERSION = 2
BLOCK = 4096
_HEADER = struct.Struct("<8sIIIIQQQQ")     # magic, version, n_samples, n_chunks, has_af, names off/len, var_count off, table off
SUFFIX = ".utm"


def _aligned(offset):
    return (offset + BLOCK - 1) // BLOCK * BLOCK


class StoreWriter:
    """with StoreWriter(path, samples, has_af) as w: w.add_chunk(n_var, columns, af32) ...; w.finish(var_count)"""

    def __init__(self, path, samples, has_af):
        self.path = path
        self.samples = [str(s) for s in samples]
        self.has_af = bool(has_af)
        self.table = []
        self.fh = open(path, "wb")
        self.fh.write(b"\0" * BLOCK)               # the header comes last
        self.done = False

    def _append(self, array):
        offset = _aligned(self.fh.tell())
        self.fh.seek(offset)
        self.fh.write(memoryview(np.ascontiguousarray(array)).cast("B"))
        return offset

    def add_chunk(self, n_var, columns, af32=None):
        """columns: uint64 (n_samples, W) as downloaded from the device; af32: float32 (n_var,) or None."""
        words = (int(n_var) + 63) // 64
        columns = np.ascontiguousarray(columns, dtype="<u8")
        if columns.shape != (len(self.samples), words):
            raise ValueError(f"chunk columns have shape {columns.shape}, expected {(len(self.samples), words)}")
        af_off = 0
        if self.has_af:
            if af32 is None or len(af32) != n_var:
                raise ValueError("an --af store needs the chunk's AF values")
            af_off = self._append(np.asarray(af32, dtype="<f4"))
        self.table.append((int(n_var), words, af_off, self._append(columns)))

    def finish(self, var_count):
        names = "\n".join(self.samples).encode("utf-8")
        names_off = self._append(np.frombuffer(names, dtype=np.uint8)) if names else _aligned(self.fh.tell())
        vc_off = self._append(np.asarray(var_count, dtype="<i8"))
        table_off = self._append(np.asarray(self.table, dtype="<u8").reshape(-1, 4))
        self.fh.seek(0)
        self.fh.write(_HEADER.pack(MAGIC, VERSION, len(self.samples), len(self.table), int(self.has_af),
                                   names_off, len(names), vc_off, table_off))
        self.fh.close()
        self.done = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if not self.done:
            self.fh.close()


class StoreReader:
    """Header, names, var_count and the chunk table of a store; column / AF blocks are mapped on request."""

    def __init__(self, path):
        self.path = path
        with open(path, "rb") as fh:
            head = fh.read(_HEADER.size)
            if len(head) < _HEADER.size:
                raise ValueError(f"{path}: not a matrix store (too short)")
            magic, version, n_samples, n_chunks, has_af, names_off, names_len, vc_off, table_off = _HEADER.unpack(head)
            if magic != MAGIC:
                raise ValueError(f"{path}: not a matrix store (incomplete write, or another format)")
            if version != VERSION:
                raise ValueError(f"{path}: store version {version}, this build reads {VERSION}: recreate it with --lowmem")
            fh.seek(names_off)
            names = fh.read(names_len).decode("utf-8")
            self.samples = np.array(names.split("\n") if names_len else [], dtype=str)
            if len(self.samples) != n_samples:
                raise ValueError(f"{path}: {len(self.samples)} sample names for {n_samples} samples")
            fh.seek(vc_off)
            self.var_count = np.frombuffer(fh.read(8 * n_samples), dtype="<i8").astype(np.int64)
            fh.seek(table_off)
            self.table = np.frombuffer(fh.read(32 * n_chunks), dtype="<u8").reshape(n_chunks, 4)
        self.has_af = bool(has_af)
        self.n_samples = int(n_samples)

    @property
    def chunk_vars(self):
        return [int(row[0]) for row in self.table]

    def columns(self, chunk, first=0, count=None):
        """Memory map of samples [first, first + count) of a chunk: uint64 (count, W).  Pages are read on access."""
        n_var, words, _af, off = (int(x) for x in self.table[chunk])
        count = self.n_samples - first if count is None else count
        return np.memmap(self.path, dtype="<u8", mode="r", offset=off + first * words * 8, shape=(count, words))

    def af(self, chunk):
        n_var, _words, af_off, _off = (int(x) for x in self.table[chunk])
        if not self.has_af or af_off == 0:
            return None
        return np.memmap(self.path, dtype="<f4", mode="r", offset=af_off, shape=(n_var,))
