"""`utmos select` on MI355X: the host side of the greedy maximum-coverage selection.

The entry points carry the names and argument meaning of the reference's module so that a caller of
utmos/select.py can switch over (what each one answers to is cited; the bodies are this build's own):

    calculate_scores(matrix, sample_mask, sample_weights)              utmos/select.py:24-53
    greedy_select(matrix, total_variant_count, select_count, ...)      utmos/select.py:69-137
    run_selection(data, select_count, subset, exclude, weights)        utmos/select.py:147-195
    load_files(in_files, lowmem, buffer, calc_af)                      utmos/select.py:241-321
    parse_sample_lists / parse_weights / parse_args / select_main      utmos/select.py:327-448

`matrix` is a device.DeviceMatrix: bit-packed sample columns resident in HBM, scored by hand-written HIP kernels
behind libutmos_hip.so.  There is no CPU path: every scoring call goes through the C ABI and raises if the
library or a GPU is missing.
"""
import argparse
import json
import logging
import os
import sys

import numpy as np

from . import device
from .store import SUFFIX as STORE_SUFFIX
from .store import StoreReader, StoreWriter

MAXMEM = 2  # GB one HBM chunk of packed columns may take; 0 = smallest chunks (the reference's test hook, select.py:18-19)

HEADER = "sample\tvar_count\tnew_count\ttot_captured\tpct_captured\n"

SELECTABLE, USED, EXCLUDED = 1, 0, 2      # sample_mask values (select.py:168: "1 = can use, 0 = mask, 2 = exclude")


# ---------------------------------------------------------------------------------------------- scoring
def calculate_scores(matrix, sample_mask, sample_weights=None):
    """Best scoring sample for the given mask: (column index, new_variant_count) or (None, None).

    Stateless like the reference: covered variants are recomputed from sample_mask == 0.  Scoring, masking,
    weighting and the first-maximum argmax all run on the device (utm_local_best); nothing is selected.  The
    greedy driver below does not call this per iteration (it keeps the whole loop on the device); it is the
    drop-in for callers that do.
    """
    sample_mask = np.asarray(sample_mask)
    matrix.set_state(sample_mask)
    matrix.set_weights(sample_weights)
    matrix.reset()
    score, use_sample, new_variant_count = matrix.local_best()
    # np.argmax runs over every sample and masked ones hold 0 (select.py:43-48): a best of 0, or a negative best
    # while some sample is masked, means "nothing to select"
    if use_sample < 0 or score == 0 or (score < 0 and (sample_mask != SELECTABLE).any()):
        return None, None
    return np.int64(use_sample), np.int64(new_variant_count)


def is_memsafe(shape, with_af=False):
    """Would a (variants, samples) block fit one HBM chunk under --maxmem?  (The reference asks this of host RAM,
    select.py:56-63; here it only decides how the variant axis is cut into chunks and never changes results.)"""
    gigabytes = (shape[0] * shape[1] / 8 + (shape[0] * 8 if with_af else 0)) / 1e9
    logging.debug("packed block of %d x %d: %.2f GB", shape[0], shape[1], gigabytes)
    return gigabytes < MAXMEM


def greedy_select(matrix, total_variant_count, select_count, vcf_samples, sample_mask, sample_weights=None, batch=64):
    """Generator of [sample, var_count, new_count, tot_captured, pct_captured], one per selected sample.

    matrix               device.DeviceMatrix; when it holds one shard of a multi-GPU run its exchange must be set up
                         (sharded.connect_shards) -- every rank then yields the same rows
    total_variant_count  per-sample totals (var_count column)
    select_count         how many samples to select at most
    vcf_samples          sample names, aligned with sample_mask
    sample_mask          1 selectable / 0 used / 2 excluded; winners are set to 0 in place, like select.py:100
    sample_weights       optional float64 per sample

    The loop itself runs on the device, `batch` iterations per call; it ends by itself when the best score is 0
    (no row for that iteration) or right after the row that captures the last variant.
    """
    n_variants = matrix.shape[0]
    if n_variants == 0:                         # nothing informative was loaded: a first pass would find only zeros
        logging.warning("No informative variants: nothing to select")
        return
    matrix.set_state(np.asarray(sample_mask))
    matrix.set_weights(sample_weights)
    matrix.reset()
    captured = 0
    wanted = int(select_count)
    while wanted > 0:
        asked = min(batch, wanted)
        winners, gains, _ = matrix.run(asked)   # up to `asked` iterations, no host round trip in between
        for winner, gain in zip(winners, gains):
            captured += gain                    # stays np.int64: the ratio below is then numpy's division and rounding
            sample_mask[winner] = USED
            yield [vcf_samples[winner], int(total_variant_count[winner]), int(gain), int(captured),
                   round(captured / n_variants, 4)]
        wanted -= len(winners)
        if len(winners) < asked:                # the device loop stopped on its own
            if captured >= n_variants:
                logging.warning("Every variant is captured: stopping early")
            else:
                logging.warning("No selectable sample adds a variant any more: stopping early")
            return


# ---------------------------------------------------------------------------------------------- set-up
def resolve_select_count(n_samples, requested):
    """--count: negative = every sample; below 1 = that fraction of the samples, at least one; else a number of
    samples (so 1 means one sample, not 100 %) -- select.py:157-159."""
    if requested < 0:
        return n_samples
    if requested < 1:
        return max(1, int(n_samples * requested))
    return max(1, int(requested))


def sample_states(names, subset, exclude):
    """uint8 state per sample: everything selectable; with a subset only its members; excluded names never."""
    states = np.full(len(names), SELECTABLE, dtype=np.uint8)
    if subset:
        states[~np.isin(names, subset)] = EXCLUDED
    if exclude:
        states[np.isin(names, exclude)] = EXCLUDED
    return states


def weight_vector(names, weights):
    """float64 weight per sample from a {name: weight} mapping (unlisted samples weigh 1), or None."""
    if weights is None:
        return None
    vector = np.ones(len(names), dtype=np.float64)
    for position, name in enumerate(names):
        weight = weights.get(name)
        if weight is not None:
            vector[position] = weight
    return vector


def run_selection(data, select_count=0.02, subset=None, exclude=None, weights=None):
    """Prepare a selection over `data` ({"data": DeviceMatrix, "samples", "var_count"}) and return its row generator."""
    n_variants, n_samples = data["data"].shape
    logging.info("%d samples x %d informative variants", n_samples, n_variants)
    rounds = resolve_select_count(n_samples, select_count)
    logging.info("up to %d samples will be selected", rounds)
    names = np.asarray(data["samples"]).astype(str)
    states = sample_states(names, subset, exclude)
    if subset or exclude:
        logging.info("subset of %d names, %d names excluded: %d samples remain selectable",
                     len(subset or ()), len(exclude or ()), int((states == SELECTABLE).sum()))
    vector = weight_vector(names, weights)
    if vector is not None:
        logging.info("%d sample weights given", len(weights))
    return greedy_select(data["data"], np.asarray(data["var_count"]), rounds, names, states, vector)


# ---------------------------------------------------------------------------------------------- ingest
def _read_part(path):
    """One input file -> {'GT': packbits rows, 'AF': per-variant float64, 'samples'} (the reference's part dict)."""
    if path.endswith((".vcf.gz", ".vcf")):
        from .vcfio import read_vcf
        return read_vcf(path)
    if path.endswith(".jl"):
        import joblib  # the reference's own container (utmos/convert.py:98)
        return joblib.load(path)
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return {"GT": z["GT"], "AF": z["AF"], "samples": z["samples"]}
    logging.error("%s: unknown file type (expected .vcf, .vcf.gz, .jl, .npz or a %s store)", path, STORE_SUFFIX)
    sys.exit(1)


def _shard_matrix(n_samples, dev, shard):
    """shard = (rank, world) or None: this process holds a contiguous block of the sample axis."""
    if shard is None or shard[1] == 1:
        return device.DeviceMatrix(n_samples, device=dev)
    from .sharded import shard_bounds
    first, n_local = shard_bounds(n_samples, *shard)
    return device.DeviceMatrix(n_samples, device=dev, first_sample=first, n_local=n_local)


def load_store(path, dev=0, shard=None):
    """Open a packed store and upload this process's columns, chunk by chunk, from a memory map."""
    try:
        reader = StoreReader(path)
    except (OSError, ValueError) as err:
        logging.error("%s", err)
        sys.exit(1)
    matrix = _shard_matrix(reader.n_samples, dev, shard)
    lo, n_local = matrix.first_sample, matrix.n_local
    for k, n_var in enumerate(reader.chunk_vars):
        chunk = matrix.add_chunk(n_var)
        matrix.upload_columns(chunk, reader.columns(k, lo, n_local))
        if reader.has_af:
            matrix.set_af(chunk, np.asarray(reader.af(k)))
    return {"samples": reader.samples, "data": matrix, "var_count": reader.var_count[lo:lo + n_local], "has_af": reader.has_af}


class _ChunkBuilder:
    """Collects consecutive input parts until they make one HBM chunk (every chunk costs its own launches in every
    iteration, so a run over many input files should not end up with one small chunk per file), uploads it, and --
    with a store being written -- streams it to disk at once, so the host never holds more than one chunk."""

    def __init__(self, matrix, writer, calc_af, af_as_f32):
        self.matrix, self.writer, self.calc_af, self.af_as_f32 = matrix, writer, calc_af, af_as_f32
        self.rows, self.af = [], []
        self.chunk_af = []                       # per flushed chunk, kept only while no AF could be set yet

    def held_variants(self):
        return sum(len(r) for r in self.rows)

    def add(self, rows, af):
        self.rows.append(rows)
        self.af.append(af)

    def flush(self):
        if not self.rows:
            return
        rows = self.rows[0] if len(self.rows) == 1 else np.concatenate(self.rows)
        af = self.af[0] if len(self.af) == 1 else np.concatenate(self.af)
        self.rows, self.af = [], []
        chunk = self.matrix.add_chunk(len(rows))
        self.matrix.upload_rows_packed(chunk, rows)          # bit transpose to columns happens on the GPU
        if self.writer is not None:
            self.writer.add_chunk(len(rows), self.matrix.download_columns(chunk), af.astype(np.float32) if self.calc_af else None)
        self.chunk_af.append(af)


def load_files(in_files, lowmem=None, buffer=32768, calc_af=False, dev=0, shard=None):
    """Load and concatenate the inputs into one HBM-resident matrix -> {"samples", "data", "var_count", "has_af"}.

    lowmem == 1: in_files[0] is an existing packed store.  lowmem == path: the store is (re)created there while
    loading; like the reference's hdf5 it keeps AF as float32, so `--lowmem` + `--af` scores with float32 AF values
    (answer_key/select_af_h5.txt) and the plain path with float64 (select_af.txt).  `buffer` = variants per chunk
    when chunking is forced (--maxmem 0).
    """
    logging.info("reading %d input file(s)", len(in_files))
    if lowmem == 1:
        return load_store(in_files[0], dev, shard)
    if shard is not None and shard[1] > 1 and lowmem is not None:
        logging.error("Create the matrix store with a single process, then select from it with several")
        sys.exit(1)

    samples = matrix = builder = writer = None
    for number, path in enumerate(in_files, start=1):
        part = _read_part(path)
        if samples is None:
            samples = np.asarray(part["samples"]).astype("U")
            matrix = _shard_matrix(len(samples), dev, shard)
            if lowmem is not None:
                writer = StoreWriter(lowmem, samples, calc_af)
            builder = _ChunkBuilder(matrix, writer, calc_af, lowmem is not None)
        rows = np.ascontiguousarray(part["GT"], dtype=np.uint8)
        informative = rows.any(axis=1)              # a variant nobody carries has no set bit in any byte (select.py:276-279)
        logging.debug("%s: %d of %d variants carried by nobody, dropped", path, int((~informative).sum()), len(rows))
        rows = rows[informative]
        af = np.asarray(part["AF"], dtype=np.float64).reshape(-1)[informative]
        if len(rows) == 0:
            continue
        if MAXMEM != 0 and is_memsafe((len(rows), len(samples)), calc_af):
            # parts are merged while the chunk stays within --maxmem
            if builder.held_variants() and not is_memsafe((builder.held_variants() + len(rows), len(samples)), calc_af):
                builder.flush()
            builder.add(rows, af)
        else:
            # a part too large for one chunk is cut into the largest pieces that fit; `buffer` variants per chunk only
            # when chunking is forced (--maxmem 0): chunks are HBM allocations here, not the I/O granularity they are
            # for the reference's hdf5 appends
            builder.flush()
            if MAXMEM != 0:
                per_variant = len(samples) / 8 + (8 if calc_af else 0)
                step = max(64, int(MAXMEM * 1e9 / per_variant) // 64 * 64 - 64)
            else:
                step = max(64, buffer // 64 * 64)
            for lo in range(0, len(rows), step):
                builder.add(rows[lo:lo + step], af[lo:lo + step])
                builder.flush()
        logging.debug("%d of %d files read", number, len(in_files))
    if builder is not None:
        builder.flush()
    if matrix is None:
        logging.error("No input files")
        sys.exit(1)

    # var_count before AF == 0 rows are cleared, like select.py:281-284.  (A shard counts its own samples;
    # select_main gathers the shards' parts.)
    var_count = matrix.var_count() if matrix.chunk_vars else np.zeros(matrix.n_local, dtype=np.int64)
    if calc_af:
        for chunk, af in enumerate(builder.chunk_af):
            matrix.set_af(chunk, af.astype(np.float32) if lowmem is not None else af)
    if writer is not None:
        writer.finish(var_count)
    return {"samples": samples, "data": matrix, "var_count": var_count, "has_af": bool(calc_af)}


# ---------------------------------------------------------------------------------------------- command line
def parse_sample_lists(argument):
    """--subset / --exclude values: each item is a file of names (one per line) or a comma separated list."""
    names = []
    for item in argument or []:
        if os.path.exists(item):
            with open(item, "r") as fh:
                names += [line.strip() for line in fh]
        else:
            names += item.split(",")
    return names


def parse_weights(argument):
    """Tab-delimited `sample<TAB>weight` -> {sample: weight}."""
    if not argument:
        return None
    table = {}
    with open(argument, "r") as fh:
        for line in fh:
            if not line.strip():
                continue
            name, weight = line.rstrip("\n").split("\t")[:2]
            table[name] = float(weight)
    return table


def setup_logging(debug=False):
    logging.basicConfig(stream=sys.stderr, level=logging.DEBUG if debug else logging.INFO,
                        format="%(asctime)s [%(levelname)s] %(message)s", force=True)


def _build_parser():
    """The option surface of the reference's `utmos select` (flags, defaults and meaning: utmos/select.py:355-398),
    plus this build's --device / --brute-force / --exchange."""
    parser = argparse.ArgumentParser(prog="select", description="Select the fewest samples that capture the most variants (MI355X).")
    parser.add_argument("in_files", nargs="*", type=str, help="inputs: .vcf[.gz], .jl, .npz parts, or one packed .utm store")
    parser.add_argument("-c", "--count", type=float, default=0.02,
                        help="how many samples: a fraction of all if < 1, a count if >= 1, every sample if -1 [%(default)s]")
    parser.add_argument("-o", "--out", type=str, default="/dev/stdout", help="TSV destination [stdout]")
    parser.add_argument("--debug", action="store_true", help="debug-level logging")
    scoring = parser.add_argument_group("Scoring Arguments")
    scoring.add_argument("--af", action="store_true", help="score variants by allele frequency instead of 1")
    scoring.add_argument("--weights", type=str, default=None, help="two-column TSV: sample, weight")
    scoring.add_argument("--subset", type=str, default=None, action="append",
                         help="only consider these samples (file or comma list; repeatable)")
    scoring.add_argument("--exclude", type=str, default=None, action="append",
                         help="never select these samples (file or comma list; repeatable)")
    memory = parser.add_argument_group("Memory Arguments")
    memory.add_argument("--lowmem", type=str, default=None,
                        help="packed matrix store (.utm) to write, or to read when no inputs are given")
    memory.add_argument("--buffer", type=int, default=32768, help="variants per HBM chunk when chunking is on [%(default)s]")
    memory.add_argument("--maxmem", type=int, default=2, help="GB one chunk may take; 0 = always chunk [%(default)s]")
    gpu = parser.add_argument_group("GPU Arguments")
    gpu.add_argument("--device", type=int, default=None, help="GPU index [0; LOCAL_RANK when launched one process per GPU]")
    gpu.add_argument("--brute-force", action="store_true",
                     help="re-score every sample from scratch in every iteration (default: later iterations only "
                          "subtract what the last winner newly captured; identical output)")
    gpu.add_argument("--exchange", choices=["auto", "mailboxes", "rccl", "rccl-allreduce"], default="auto",
                     help="several processes: how the shards meet every iteration [auto: device mailboxes, else RCCL]")
    return parser


def parse_args(args):
    """Parse and validate the command line; exits with status 1 on unusable input combinations (select.py:401-415)."""
    args = _build_parser().parse_args(args)
    setup_logging(args.debug)
    stores = [name for name in args.in_files if name.endswith((STORE_SUFFIX, ".hdf5"))]
    if any(name.endswith(".hdf5") for name in args.in_files) or (args.lowmem or "").endswith(".hdf5"):
        logging.error("hdf5 stores are not read by this build; recreate the store with --lowmem FILE%s", STORE_SUFFIX)
        sys.exit(1)
    if stores and len(args.in_files) > 1:
        logging.error("A matrix store cannot be combined with other input files")
        sys.exit(1)
    if not args.in_files and not args.lowmem:
        logging.error("Nothing to read: give input files, or --lowmem STORE to select from an existing store")
        sys.exit(1)
    if not args.in_files:                        # --lowmem STORE alone: select from the existing store
        args.in_files, args.lowmem = [args.lowmem], 1
    elif stores and not args.lowmem:
        logging.info("Input is a matrix store: reading it directly")
        args.lowmem = 1
    logging.info("options in effect: %s", json.dumps(vars(args), sort_keys=True))
    return args


def _gather_var_count(transport, local_counts, n_samples):
    """Every sample's var_count for the output rows: each shard popcounted its own columns on its GPU."""
    from .sharded import shard_bounds
    width = (n_samples + transport.world - 1) // transport.world      # shards differ by at most one sample
    padded = np.full(width, -1, dtype=np.int64)
    padded[:len(local_counts)] = local_counts
    parts = transport.allgather_bytes(padded.tobytes())
    return np.concatenate([np.frombuffer(blob, dtype=np.int64)[:shard_bounds(n_samples, rank, transport.world)[1]]
                           for rank, blob in enumerate(parts)])


def select_main(cmdargs):
    """`utmos select`: load, select, write the TSV (header + one flushed line per selected sample, select.py:440-446)."""
    global MAXMEM
    args = parse_args(cmdargs)
    MAXMEM = args.maxmem
    missing = [path for path in args.in_files if not os.path.exists(path)]
    if missing:
        logging.error("Input %s does not exist", missing[0])
        sys.exit(1)

    # one process per GPU (torchrun-style RANK / WORLD_SIZE / LOCAL_RANK): each holds a block of the samples
    from .sharded import bootstrap, connect_shards, dist_env
    rank, world, local_rank = dist_env()
    if args.device is not None:
        dev = args.device
    else:   # one GPU per rank on a full node (ranks share devices only on smaller test boxes)
        dev = local_rank % device.nat.device_count() if world > 1 else 0
    data = load_files(args.in_files, args.lowmem, args.buffer, args.af, dev, (rank, world) if world > 1 else None)
    matrix = data["data"]
    if not args.brute_force:
        matrix.set_decremental(True)            # exact; same rows (DESIGN.md "Decremental scoring")
    matrix.set_af_exact_scores(False)           # the TSV has no score column: only ambiguous argmaxes need their chains
    if not data["has_af"] and args.af:
        logging.critical("The store was created without --af: remove --af or recreate it")
        sys.exit(1)
    if data["has_af"] and not args.af:
        logging.critical("The store was created with --af and scores by allele frequency: add --af or recreate it")

    transport = None
    if world > 1:
        transport, uid = bootstrap(rank, world, device.DeviceMatrix.comm_unique_id)
        data["var_count"] = _gather_var_count(transport, data["var_count"], len(data["samples"]))
        try:
            how = connect_shards(matrix, transport, uid, os.environ.get("UTMOS_EXCHANGE", args.exchange))
        except RuntimeError as err:
            logging.critical("%s", err)
            sys.exit(1)
        logging.info("rank %d of %d: per-iteration exchange through %s", rank, world, how)
        if rank != 0:
            args.out = os.devnull

    subset = parse_sample_lists(args.subset)
    exclude = parse_sample_lists(args.exclude)
    weights = parse_weights(args.weights)
    try:
        with open(args.out, "w") as fout:
            fout.write(HEADER)
            for row in run_selection(data, args.count, subset, exclude, weights):
                logging.info("%s selected: %.1f%% of the variants captured", row[0], row[4] * 100)
                fout.write("\t".join(str(field) for field in row) + "\n")
                fout.flush()
    finally:
        if transport is not None:
            # nobody unmaps or frees its columns while a peer's last launches may still read them
            try:
                transport.barrier()
            except OSError:
                pass
            transport.close()
        matrix.close()
    logging.info("utmos select finished")
