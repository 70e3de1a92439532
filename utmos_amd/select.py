"""`utmos select` on MI355X: the host side of the greedy maximum-coverage selection.

Mirrors the reference's module utmos/select.py function for function -- same names, argument
meaning and stopping behaviour -- with the matrix living in HBM (device.DeviceMatrix) and the loop
running in hand-written HIP kernels behind libutmos_hip.so:

    calculate_scores   utmos/select.py:24-53    one scoring pass, stateless in sample_mask
    greedy_select      utmos/select.py:69-137   generator of [name, var_count, new, tot, pct]
    run_selection      utmos/select.py:147-195
    load_files         utmos/select.py:241-321  (in-memory branch; the hdf5 branch -> packed .utm store)
    parse_sample_lists / parse_weights / parse_args / select_main   utmos/select.py:327-448

There is no CPU path: every scoring call goes through the C ABI and raises if the library or a GPU
is missing.
"""
import argparse
import json
import logging
import os
import sys

import numpy as np

from . import device

MAXMEM = 2  # GB of HBM a chunk of packed columns may take; 0 forces the smallest chunks (test hook, as select.py:18-19)

HEADER = "sample\tvar_count\tnew_count\ttot_captured\tpct_captured\n"
STORE_SUFFIX = ".utm"


#############
# Core code #
#############
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
    if use_sample < 0 or score == 0 or (score < 0 and (sample_mask != 1).any()):
        return None, None
    return np.int64(use_sample), np.int64(new_variant_count)


def is_memsafe(shape, with_af=False):
    """HBM estimate in GB of the packed matrix (+ AF table); True if it fits one chunk."""
    data_size = (shape[0] * shape[1] / 8 + (shape[0] * 8 if with_af else 0)) / 1e9
    logging.debug("Estimated packed size %.2fGB", data_size)
    return data_size < MAXMEM


##############
# Algorithms #
##############
def greedy_select(matrix, total_variant_count, select_count, vcf_samples, sample_mask, sample_weights=None,
                  batch=64, transport=None):
    """Greedy calculation; yields each selected sample's row.

    matrix:              device.DeviceMatrix (bit-packed, HBM resident)
    total_variant_count: total number of variants per sample
    select_count:        how many samples to select
    vcf_samples:         sample names, lines up with sample_mask
    sample_mask:         1 = can be selected, 0 = used, 2 = excluded; updated in place like the reference
    sample_weights:      optional per-sample weights
    transport:           host-staged exchange between shards (sharded.SocketTransport / TorchDistTransport);
                         None = single GPU, or RCCL already initialised on the matrix (run() is then collective)
    """
    num_vars = matrix.shape[0]
    tot_captured = 0
    if num_vars == 0:      # nothing informative was loaded: the reference's first scoring pass finds only zeros
        logging.warning("Ran out of new variants (multi-allelics)")
        return
    matrix.set_state(np.asarray(sample_mask))
    matrix.set_weights(sample_weights)
    matrix.reset()
    remaining = int(select_count)
    staged = None
    if transport is not None:
        from .sharded import sharded_greedy
        staged = sharded_greedy(matrix, transport, remaining)
    while remaining > 0:
        want = min(batch, remaining)
        if staged is None:
            idx, new, _ = matrix.run(want)      # device-resident: up to `want` iterations, no host round trips
        else:
            rows = [row for _, row in zip(range(want), staged)]
            idx, new = [r[0] for r in rows], [np.int64(r[1]) for r in rows]
        for use_sample, new_variant_count in zip(idx, new):
            tot_captured += new_variant_count   # np.int64, as in the reference
            sample_mask[use_sample] = 0
            yield [vcf_samples[use_sample], int(total_variant_count[use_sample]), int(new_variant_count),
                   int(tot_captured), round(tot_captured / num_vars, 4)]
        remaining -= len(idx)
        if len(idx) < want:                     # the device loop stopped by itself
            if tot_captured >= num_vars:
                logging.warning("Ran out of new variants")
            else:
                logging.warning("Ran out of new variants (multi-allelics)")
            return


####################
# Setup/Management #
####################
def run_selection(data, select_count=0.02, subset=None, exclude=None, weights=None, transport=None):
    """Set up the selection: select_count in [0,1) = fraction, >= 1 = count, < 0 = all."""
    num_vars, num_samples = data["data"].shape
    logging.info("Sample Count %d", num_samples)
    logging.info("Variant Count %d", num_vars)

    select_count = num_samples if select_count < 0 \
        else max(1, int(num_samples * select_count) if select_count < 1 else int(select_count))
    logging.info("Selecting %d samples", select_count)

    vcf_samples = np.asarray(data["samples"]).astype(str)

    # 1 = can use, 0 = mask, 2 = exclude
    sample_mask = np.ones(num_samples, dtype="uint8")
    if subset:
        sample_mask = np.where(np.isin(vcf_samples, subset), 1, 2).astype("uint8")
        logging.info("Subsetting to %d samples", len(subset))
    if exclude:
        sample_mask = np.where(np.isin(vcf_samples, exclude), 2, sample_mask).astype("uint8")
        logging.info("Excluding %d samples", len(exclude))
    if subset and exclude:
        logging.info("Ending with %d samples", len(sample_mask) - (sample_mask == 1).sum())

    sample_weights = None
    if weights is not None:
        logging.info("Setting %d weights", len(weights))
        sample_weights = np.ones(num_samples)
        for pos, name in enumerate(vcf_samples):
            if name in weights:
                sample_weights[pos] = weights[name]

    return greedy_select(data["data"], np.asarray(data["var_count"]), select_count, vcf_samples, sample_mask,
                         sample_weights, transport=transport)


def _read_part(path):
    if path.endswith((".vcf.gz", ".vcf")):
        from .vcfio import read_vcf
        return read_vcf(path)
    if path.endswith(".jl"):
        import joblib  # the reference's own container (utmos/convert.py:98)
        return joblib.load(path)
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return {"GT": z["GT"], "AF": z["AF"], "samples": z["samples"]}
    logging.error("Unknown filetype %s. Expected `.vcf[.gz]`, `.jl`, `.npz`", path)
    sys.exit(1)


def save_store(path, data, af_parts):
    """Packed column store (the hdf5 replacement): samples, var_count and per chunk the column bitsets
    (+ float32 AF, the dtype the reference's hdf5 holds, select.py:218-223)."""
    matrix = data["data"]
    arrays = {"samples": np.asarray(data["samples"]).astype("U"), "var_count": np.asarray(data["var_count"]),
              "chunk_vars": np.asarray(matrix.chunk_vars, dtype=np.int64), "has_af": np.asarray(af_parts is not None)}
    for c in range(len(matrix.chunk_vars)):
        arrays[f"cols{c}"] = matrix.download_columns(c)
        if af_parts is not None:
            arrays[f"af{c}"] = np.asarray(af_parts[c], dtype=np.float32)
    with open(path, "wb") as fh:
        np.savez(fh, **arrays)


def _shard_matrix(n_samples, dev, shard):
    """shard = (rank, world) or None: this process holds a contiguous block of the sample axis."""
    if shard is None or shard[1] == 1:
        return device.DeviceMatrix(n_samples, device=dev)
    from .sharded import shard_bounds
    first, n_local = shard_bounds(n_samples, *shard)
    return device.DeviceMatrix(n_samples, device=dev, first_sample=first, n_local=n_local)


def load_store(path, dev=0, shard=None):
    with np.load(path, allow_pickle=False) as z:
        samples = z["samples"]
        matrix = _shard_matrix(len(samples), dev, shard)
        lo, hi = matrix.first_sample, matrix.first_sample + matrix.n_local
        has_af = bool(z["has_af"])
        for c, n_var in enumerate(z["chunk_vars"]):
            idx = matrix.add_chunk(int(n_var))
            matrix.upload_columns(idx, z[f"cols{c}"][lo:hi])
            if has_af:
                matrix.set_af(idx, z[f"af{c}"])
        return {"samples": samples, "data": matrix, "var_count": z["var_count"][lo:hi], "has_af": has_af}


# pylint: disable=too-many-locals
def load_files(in_files, lowmem=None, buffer=32768, calc_af=False, dev=0, shard=None):
    """Load and concatenate inputs into one HBM-resident matrix.

    lowmem == 1: in_files[0] is an existing packed store.  lowmem == path: the store is (re)created
    there; like the reference's hdf5 it keeps AF as float32, so `--lowmem` + `--af` scores with
    float32 AF values (answer_key/select_af_h5.txt), the plain path with float64 (select_af.txt).
    `buffer` = variants per chunk when chunking is forced.
    """
    logging.info("Loading %d files", len(in_files))
    if lowmem == 1:
        return load_store(in_files[0], dev, shard)
    sharded = shard is not None and shard[1] > 1
    if sharded and lowmem is not None:
        logging.error("Create the matrix store with a single process, then select from it with several")
        sys.exit(1)

    samples = None
    matrix = None
    af_parts = []
    held_rows, held_af = [], []          # consecutive parts waiting to become one chunk

    def flush():
        """The held parts become one chunk: every chunk costs its own launches in every iteration, so a run over
        many input files should not end up with one small chunk per file."""
        if not held_rows:
            return
        rows = held_rows[0] if len(held_rows) == 1 else np.concatenate(held_rows)
        chunk = matrix.add_chunk(len(rows))
        matrix.upload_rows_packed(chunk, rows)
        af_parts.append(held_af[0] if len(held_af) == 1 else np.concatenate(held_af))
        held_rows.clear()
        held_af.clear()

    for load_count, path in enumerate(in_files):
        dat = _read_part(path)
        if samples is None:
            samples = np.asarray(dat["samples"]).astype("U")
            matrix = _shard_matrix(len(samples), dev, shard)
        rows = np.ascontiguousarray(dat["GT"], dtype=np.uint8)
        informative = rows.any(axis=1)          # a row without carriers has no set bit in any byte
        logging.debug("fitering %d uninformative variants", int((~informative).sum()))
        rows = rows[informative]
        af = np.asarray(dat["AF"], dtype=np.float64).reshape(-1)[informative]
        if len(rows) == 0:
            continue                            # a part without a single carrier contributes nothing
        if MAXMEM != 0 and is_memsafe((len(rows), len(samples)), calc_af):
            # parts are merged while the chunk stays within --maxmem (is_memsafe: the policy never changes results)
            held = sum(len(r) for r in held_rows)
            if held and not is_memsafe((held + len(rows), len(samples)), calc_af):
                flush()
            held_rows.append(rows)
            held_af.append(af)
        else:
            # a part too large for one chunk is cut into the largest memsafe pieces; `buffer` variants per chunk only
            # when chunking is forced (--maxmem 0, the reference's test hook): chunks are HBM allocations here, not
            # the I/O granularity they are for the reference's hdf5 appends, and each one costs launches per iteration
            flush()
            if MAXMEM != 0:
                per_variant = len(samples) / 8 + (8 if calc_af else 0)
                step = max(64, int(MAXMEM * 1e9 / per_variant) // 64 * 64 - 64)
            else:
                step = max(64, buffer // 64 * 64)
            for lo in range(0, len(rows), step):
                held_rows.append(rows[lo:lo + step])
                held_af.append(af[lo:lo + step])
                flush()
        logging.debug("Loaded %d of %d", load_count + 1, len(in_files))
    flush()

    ret = {"samples": samples, "data": matrix}
    # before AF == 0 rows are cleared, like select.py:281-284.  (A shard counts its own samples; select_main
    # gathers the shards' parts.)
    ret["var_count"] = matrix.var_count() if matrix.chunk_vars else np.zeros(matrix.n_local, dtype=np.int64)
    if calc_af:
        as32 = lowmem is not None
        for chunk, af in enumerate(af_parts):
            matrix.set_af(chunk, af.astype(np.float32) if as32 else af)
    ret["has_af"] = bool(calc_af)
    if lowmem is not None:
        save_store(lowmem, ret, af_parts if calc_af else None)
    return ret
# pylint: enable=too-many-locals


###################
# Input utilities #
###################
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
    ret = {}
    with open(argument, "r") as fh:
        for line in fh:
            if not line.strip():
                continue
            name, weight = line.rstrip("\n").split("\t")[:2]
            ret[name] = float(weight)
    return ret


def setup_logging(debug=False):
    logging.basicConfig(stream=sys.stderr, level=logging.DEBUG if debug else logging.INFO,
                        format="%(asctime)s [%(levelname)s] %(message)s", force=True)


# (flags, keyword arguments) -- the option surface of the reference's `utmos select` (utmos/select.py:355-398)
_GENERAL = [
    (("in_files",), dict(nargs="*", type=str, help="inputs: .vcf[.gz], .jl, .npz parts, or one packed .utm store")),
    (("-c", "--count"), dict(type=float, default=0.02,
                            help="how many samples: a fraction of all if < 1, a count if >= 1, every sample if -1 [%(default)s]")),
    (("-o", "--out"), dict(type=str, default="/dev/stdout", help="TSV destination [stdout]")),
    (("--debug",), dict(action="store_true", help="debug-level logging")),
]
_SCORING = [
    (("--af",), dict(action="store_true", help="score variants by allele frequency instead of 1")),
    (("--weights",), dict(type=str, default=None, help="two-column TSV: sample, weight")),
    (("--subset",), dict(type=str, default=None, action="append", help="only consider these samples (file or comma list; repeatable)")),
    (("--exclude",), dict(type=str, default=None, action="append", help="never select these samples (file or comma list; repeatable)")),
]
_MEMORY = [
    (("--lowmem",), dict(type=str, default=None, help="packed matrix store (.utm) to write, or to read when no inputs are given")),
    (("--buffer",), dict(type=int, default=32768, help="variants per HBM chunk when chunking is on [%(default)s]")),
    (("--maxmem",), dict(type=int, default=2, help="GB one chunk may take; 0 = always chunk [%(default)s]")),
    (("--device",), dict(type=int, default=0, help="GPU index [%(default)s]")),
    (("--brute-force",), dict(action="store_true",
                             help="re-score every sample from scratch in every iteration (default: later iterations only "
                                  "subtract what the last winner newly captured; identical output)")),
]


def parse_args(args):
    """Parse and validate the command line; exits with status 1 on unusable input combinations."""
    parser = argparse.ArgumentParser(prog="select", description="Select the fewest samples that capture the most variants (MI355X).")
    for flags, kw in _GENERAL:
        parser.add_argument(*flags, **kw)
    for title, table in (("Scoring Arguments", _SCORING), ("Memory Arguments", _MEMORY)):
        group = parser.add_argument_group(title)
        for flags, kw in table:
            group.add_argument(*flags, **kw)
    args = parser.parse_args(args)
    setup_logging(args.debug)

    stores = [f for f in args.in_files if f.endswith((STORE_SUFFIX, ".hdf5"))]
    if stores and len(args.in_files) > 1:
        logging.error("A matrix store cannot be combined with other input files")
        sys.exit(1)
    if any(f.endswith(".hdf5") for f in args.in_files) or (args.lowmem or "").endswith(".hdf5"):
        logging.error("hdf5 stores are not read by this build; recreate with --lowmem FILE%s", STORE_SUFFIX)
        sys.exit(1)
    if not args.in_files:
        if not args.lowmem:
            logging.error("No input files provided")
            sys.exit(1)
        args.in_files, args.lowmem = [args.lowmem], 1      # reuse an existing store
    elif stores and not args.lowmem:
        logging.info("Input is a matrix store: reading it directly")
        args.lowmem = 1
    logging.info("Params:\n%s", json.dumps(vars(args), indent=4))
    return args


def _padded(local_counts, world, n_samples):
    """Shards differ by at most one sample: pad to a common length for the fixed-size exchange."""
    width = (n_samples + world - 1) // world
    out = np.full(width, -1, dtype=np.int64)
    out[:len(local_counts)] = local_counts
    return out


def _unpadded(parts, world, n_samples):
    from .sharded import shard_bounds
    chunks = []
    for rank, blob in enumerate(parts):
        n_local = shard_bounds(n_samples, rank, world)[1]
        chunks.append(np.frombuffer(blob, dtype=np.int64)[:n_local])
    return np.concatenate(chunks)


def select_main(cmdargs):
    """Main"""
    global MAXMEM  # pylint: disable=global-statement
    args = parse_args(cmdargs)
    MAXMEM = args.maxmem
    for path in args.in_files:
        if not os.path.exists(path):
            logging.error("Input %s does not exist", path)
            sys.exit(1)

    # one process per GPU (torchrun-style RANK / WORLD_SIZE / LOCAL_RANK): each holds a block of the samples
    from .sharded import bootstrap, dist_env, enable_p2p
    rank, world, local_rank = dist_env()
    shard = (rank, world) if world > 1 else None
    # one GPU per rank on a full node (ranks share devices only on smaller test boxes)
    dev = local_rank % device.nat.device_count() if world > 1 and "--device" not in cmdargs else args.device
    data = load_files(args.in_files, args.lowmem, args.buffer, args.af, dev, shard)
    if not args.brute_force:
        data["data"].set_decremental(True)      # exact; same rows (DESIGN.md "Decremental scoring")
    data["data"].set_af_exact_scores(False)     # the TSV has no score column: only ambiguous argmaxes need their chains
    transport = None
    if world > 1:
        matrix = data["data"]
        host_only = os.environ.get("UTMOS_TRANSPORT", "rccl") == "socket"     # never touch RCCL (tests, hosts without it)
        transport, uid = bootstrap(rank, world, None if host_only else device.DeviceMatrix.comm_unique_id)
        # var_count of every sample, for the output rows: each shard popcounts its own columns on its GPU
        parts = transport.allgather_bytes(np.ascontiguousarray(_padded(data["var_count"], world, len(data["samples"]))).tobytes())
        data["var_count"] = _unpadded(parts, world, len(data["samples"]))
        if os.environ.get("UTMOS_P2P", "1") == "1":
            enable_p2p(matrix, transport)           # hipIpc column mappings + record mailboxes, self-tested
        if not matrix.fused and not host_only:
            try:
                matrix.comm_init(rank, world, uid)  # RCCL carries the per-iteration exchange instead
                up = 1
            except device.nat.NativeError as err:
                logging.warning("no RCCL communicator on rank %d (%s)", rank, err)
                up = 0
            ups = [r[1] for r in transport.allgather((0.0, up, 0))]
            if any(ups) and not all(ups):
                logging.critical("RCCL came up on some shards only")
                sys.exit(1)
            # nowhere: the shards keep exchanging records (and, without mappings, columns) through the host sockets
        if matrix.fused:                            # the loop runs on the devices: nothing goes through the host
            transport.close()
            transport = None
        if rank != 0:
            args.out = os.devnull
    if not data["has_af"] and args.af:
        logging.critical("Store doesn't appear to be created with --af weighted scores, remove --af or recreate it")
        sys.exit(1)
    if data["has_af"] and not args.af:
        logging.critical("Store appears to be created with --af weighted scores, add --af or recreate it")

    args.subset = parse_sample_lists(args.subset)
    args.exclude = parse_sample_lists(args.exclude)
    args.weights = parse_weights(args.weights)

    with open(args.out, "w") as fout:
        fout.write(HEADER)
        for result in run_selection(data, args.count, args.subset, args.exclude, args.weights, transport):
            logging.info("Selected %s (%.1f%% of variants)", result[0], result[4] * 100)
            fout.write("\t".join([str(_) for _ in result]) + "\n")
            fout.flush()
    if transport is not None:
        transport.close()
    data["data"].close()
    logging.info("Finished utmos")
