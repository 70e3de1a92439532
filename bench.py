#!/usr/bin/env python3
"""bench.py -- greedy iterations/sec + achieved HBM GB/s of the selection loop on MI355X.

A *step* is one full `select all` run of the greedy loop (utmos/select.py:69-112) over the synthetic
10M-variant x 2,504-sample bit matrix (BASELINE.json configs[1]), matrix already resident in HBM.
`value` = greedy iterations per second over the K timed steps (all ranks, max time over ranks).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W          # one rank per GPU; samples sharded; RCCL exchange

With N > 1 the same 10M x 2,504 problem is sharded over the sample axis (strong scaling): every
iteration ends with one ncclAllGather of {best record, best column} per rank.  No torch is imported:
ranks find each other through RANK/LOCAL_RANK/WORLD_SIZE and a rendezvous file for the ncclUniqueId.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_START = time.time()
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


# BASELINE.json `configs`, by name.  cfg2 is the default (the metric's configuration); cfg4 is meant for --gpus 8
# (12,500 columns = 78 GB per rank) and cfg5 for one GPU (156 GB in ten chunks); both select a fixed number of
# samples instead of all of them.
WORKLOADS = {
    "cfg1": ("1.1M x 2,504 (chr22-sized), select all", dict(n_var=1_103_547, n_samp=2504, select=-1)),
    "cfg2": ("10M x 2,504, select all", dict(n_var=10_000_000, n_samp=2504, select=-1)),
    "cfg3": ("10M x 2,504 with float32 AF weighting, select all", dict(n_var=10_000_000, n_samp=2504, select=-1, af=True)),
    "cfg4": ("50M x 100,000 over the ranks, first 20 iterations", dict(n_var=50_000_000, n_samp=100_000, select=20)),
    "cfg5": ("500M x 2,504 in chunks of 50M, first 10 iterations",
             dict(n_var=500_000_000, n_samp=2504, select=10, chunk_vars=50_000_000)),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--n-var", type=int, default=10_000_000)
    p.add_argument("--n-samp", type=int, default=2504)
    p.add_argument("--select", type=int, default=-1, help="samples to select per step (-1 = all)")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--af", action="store_true", help="configs[2]: float32 AF weighting")
    p.add_argument("--af-dtype", choices=["f32", "f64"], default="f32",
                   help="f32 = the reference's hdf5 values (configs[2]); f64 = its in-memory values")
    p.add_argument("--chunk-vars", type=int, default=0, help="split the variant axis into chunks of this many variants")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample-vars", type=int, default=1_000_000)
    p.add_argument("--no-roofline-pass", action="store_true")
    p.add_argument("--af-estimate-scores", action="store_true",
                   help="AF: do not chain unambiguous winners (same rows; reported scores are estimates)")
    p.add_argument("--decremental", action="store_true",
                   help="SURVEY 8f-4 shortcut (exact, reads far fewer bytes): reported separately, never the default")
    p.add_argument("--no-sharded-check", action="store_true", help="N > 1: skip rank 0's single-GPU re-run and comparison")
    p.add_argument("--exchange", choices=["auto", "rccl"], default="auto",
                   help="N > 1: auto = device mailboxes over hipIpc mappings when every rank can, else RCCL; rccl = force RCCL")
    p.add_argument("--decr-threshold", type=float, default=0.0, help="--decremental: newly-covered word fraction below which an iteration goes decremental (0 = library default)")
    p.add_argument("--pmc-traffic", choices=["live", "recorded", "off"], default="live",
                   help="roofline.traffic: live = two rocprofv3 --pmc child runs of one step of this workload (N = 1), "
                        "falling back to the passes recorded under profiles/; recorded = only those; off = null")
    p.add_argument("--workload", choices=sorted(WORKLOADS), default=None,
                   help="a BASELINE.json configuration by name (sets the shape flags): " +
                        "; ".join(f"{k}: {v[0]}" for k, v in sorted(WORKLOADS.items())))
    p.add_argument("--force-comm", action="store_true", help="initialise RCCL even with one rank (exercises the exchange path)")
    args = p.parse_args()
    if args.workload:
        for key, value in WORKLOADS[args.workload][1].items():
            setattr(args, key, value)
    return args


def live_pmc_traffic(args):
    """HBM bytes per scoring launch from the PMC counters, as MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE and
    WRITE_SIZE in separate `rocprofv3 --pmc` passes with nothing else enabled (KiB units; on gfx950 FETCH_SIZE reports
    half of a wide coalesced read -> x2).  Each pass is a child process running ONE step of this same workload.
    -> (mean bytes per launch, launches, note) or None when the profiler is not usable here."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None      # this process is itself being profiled: no nested profiler runs
    work = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline-pass", "--pmc-traffic", "off",
            "--n-var", str(args.n_var), "--n-samp", str(args.n_samp), "--select", str(args.select), "--seed", str(args.seed),
            "--chunk-vars", str(args.chunk_vars)]
    if args.af:
        work += ["--af", "--af-dtype", args.af_dtype]
    if args.af_estimate_scores:
        work += ["--af-estimate-scores"]
    tmp = os.environ.get("TMPDIR", "/tmp")
    per_launch = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out_dir = tempfile.mkdtemp(prefix="utm_pmc_", dir=tmp)
        try:
            run = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", out_dir, "--",
                                  sys.executable, os.path.abspath(__file__)] + work,
                                 cwd=tmp, env=dict(os.environ, TMPDIR=tmp), capture_output=True, text=True, timeout=600)
            files = glob.glob(os.path.join(out_dir, "**", "*_counter_collection.csv"), recursive=True)
            if run.returncode != 0 or not files:
                return None
            with open(max(files, key=os.path.getmtime)) as fh:
                per_launch[counter] = [float(r["Counter_Value"]) for r in csv.DictReader(fh)
                                       if "k_score_" in r["Kernel_Name"] and r["Counter_Name"] == counter]
        except (OSError, subprocess.SubprocessError, KeyError, ValueError):
            return None
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    n = min(len(per_launch["FETCH_SIZE"]), len(per_launch["WRITE_SIZE"]))
    if n == 0:
        return None
    total = sum(per_launch["FETCH_SIZE"][:n]) * 1024 * 2 + sum(per_launch["WRITE_SIZE"][:n]) * 1024
    return total / n, n, ("live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child runs of one step of this workload; "
                          "read bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950), write bytes = WRITE_SIZE KiB x 1024")


def rendezvous_id(rank, world, make_id):
    from utmos_amd.sharded import rendezvous_unique_id
    return rendezvous_unique_id(rank, make_id)


def cpu_baseline(args, device_mod):
    """The reference's algorithm on the host, on a bounded sample of the same synthetic workload:
    oracle.score_rowloop = the per-row numpy loop of utmos/select.py:33-48, single thread, first 3
    iterations on the first `cpu_sample_vars` variants; rate scaled to the full variant count
    (time per iteration is linear in rows).  Second line: the packed C oracle with OpenMP."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_util as ou
    n_var = min(args.cpu_sample_vars, args.n_var)
    with device_mod.DeviceMatrix(args.n_samp, device=0) as m:
        c = m.add_chunk(n_var)
        m.synth_fill(c, seed=args.seed)
        cols = m.download_columns(c)
    bits = np.unpackbits(cols.view(np.uint8), axis=1, bitorder="little")[:, :n_var]
    dense = np.ascontiguousarray(bits.T).astype(bool)
    del bits
    state = np.ones(args.n_samp, np.uint8)
    iters = 3
    t0 = time.perf_counter()
    for _ in range(iters):
        best, _new = ou.npo.score_rowloop(dense, state)
        state[best] = 0
    dt = time.perf_counter() - t0
    scale = n_var / args.n_var
    port = {"value": iters / dt * scale, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"first {iters} greedy iterations on the first {n_var} of {args.n_var} synthetic variants x "
                      f"{args.n_samp} samples (numpy row loop, {dt:.1f} s); rate scaled by {scale:.4g} to the full variant count",
            "host_cpus": os.cpu_count()}
    threads = min(os.cpu_count() or 1, 16)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    t0 = time.perf_counter()
    k = 40
    idx, _, _ = ou.c_greedy(cols, n_var, np.ones(args.n_samp, np.uint8), k_max=k, omp=True)
    dt2 = time.perf_counter() - t0
    bitset = {"value": len(idx) / dt2 * scale, "unit": "iterations/s", "cores": threads, "kind": "port",
              "sample": f"first {len(idx)} iterations, packed C bitset oracle with OpenMP, same sample, scaled the same way"}
    return port, bitset


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 needs one process per GPU: launch with python -m torch.distributed.run "
                             "--nproc-per-node N (see module docstring)")
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1:
        # a rank that lost its peers must not sit in a collective forever: say where it was and leave
        import faulthandler
        faulthandler.dump_traceback_later(900, exit=True)

    import numpy as np
    from utmos_amd import device

    n_total = args.n_samp
    first = rank * n_total // world
    n_local = (rank + 1) * n_total // world - first
    n_dev = device.nat.device_count()
    dev_index = local_rank % n_dev        # one GPU per rank on a full node; ranks share devices only on smaller test boxes
    m = device.DeviceMatrix(n_total, device=dev_index, first_sample=first, n_local=n_local)
    chunk_vars = args.chunk_vars or args.n_var
    v0 = 0
    t_gen = time.perf_counter()
    while v0 < args.n_var:
        nv = min(chunk_vars, args.n_var - v0)
        c = m.add_chunk(nv)
        m.synth_fill(c, seed=args.seed, first_var_global=v0)
        if args.af:
            _, af = device.synth_host(args.seed, nv, n_total, first_var_global=v0, want_cols=False)
            # f64: full 53-bit mantissas, like the reference's ac/an quotients
            m.set_af(c, af if args.af_dtype == "f32" else af.astype(np.float64) / 3.0)
        v0 += nv
    t_gen = time.perf_counter() - t_gen
    exchange = "none"
    host_staged = False
    id_path = None
    transport = None
    if world > 1:
        # start-up over a local TCP socket (port published in the launch's rendezvous file).  Default exchange:
        # every rank maps every other rank's columns and record mailboxes (hipIpc), the mappings are self-tested,
        # and the loop then runs without any host or collective in it.  If that is not possible on every rank
        # (or with --exchange rccl) the RCCL communicator carries the per-iteration exchange instead.
        from utmos_amd.sharded import bootstrap, enable_p2p
        transport, uid = bootstrap(rank, world, device.DeviceMatrix.comm_unique_id)
        if args.exchange != "rccl" and os.environ.get("UTM_NO_P2P", "0") == "0":
            enable_p2p(m, transport)
        if not m.fused_mailboxes:
            try:
                m.comm_init(rank, world, uid)
                ok = 1
            except device.nat.NativeError as e:
                sys.stderr.write(f"bench.py rank {rank}: RCCL communicator unavailable ({e})\n")
                ok = 0
            oks = [r[1] for r in transport.allgather((0.0, ok, 0))]
            if not all(oks):
                if any(oks):
                    raise SystemExit("bench.py: RCCL came up on some ranks only")
                host_staged = True      # last resort: records and winner columns through the host sockets
        replica = m.stats()["p2p_replica_bytes"]
        columns = (f"winner columns read from a one-time local copy of the other shards' columns ({replica / 1e9:.2f} GB)"
                   if replica else "in-place column reads over hipIpc mappings (xGMI)")
        boxes = "host shared-memory mailboxes" if getattr(m, "host_mailboxes", False) else "device mailboxes"
        exchange = f"{boxes} + {columns}" if m.fused_mailboxes else (
            "host-staged: records over TCP, winner column " + (columns if m.p2p else "through host memory")
            if host_staged else
            f"ncclAllGather of records, {columns}" if m.p2p else "ncclAllGather of records + columns")
    elif args.force_comm:
        uid, id_path = rendezvous_id(rank, world, device.DeviceMatrix.comm_unique_id)
        m.comm_init(rank, world, uid)
        exchange = "ncclAllGather (single rank)"

    def sync_max(value):
        """Barrier + maximum over the ranks (host side: the ranks' loops are already drained)."""
        if transport is None:
            return value
        return max(r[0] for r in transport.allgather((float(value), 0, 0)))

    k_sel = n_total if args.select < 0 else min(args.select, n_total)
    if args.decremental:
        m.set_decremental(True, args.decr_threshold)
    if args.af_estimate_scores:
        m.set_af_exact_scores(False)

    def one_step():
        m.reset()
        if world > 1 and host_staged:
            from utmos_amd.sharded import sharded_greedy
            rows = list(sharded_greedy(m, transport, k_sel))
            return (np.array([r[0] for r in rows], np.int64), np.array([r[1] for r in rows], np.int64),
                    np.array([r[2] for r in rows], np.float64))
        return m.run(k_sel)

    for _ in range(args.warmup):
        one_step()
    sync_max(0.0)  # barrier (run() returns only after its stream has drained)
    t0 = time.perf_counter()
    iters = 0
    loop_ms = 0.0
    for _ in range(args.steps):
        idx, new, _ = one_step()
        iters += len(idx)
        loop_ms += m.stats()["loop_ms"]
    elapsed = time.perf_counter() - t0
    elapsed = sync_max(elapsed)
    st = m.stats()
    algo_bytes_step = st["algo_bytes"]          # since the last reset = one step, this rank's shard
    tot_captured = st["tot_captured"]

    roofline = None
    if not args.no_roofline_pass and not args.decremental:   # the roofline object describes the brute-force kernel only
        # same step once more with every scoring launch bracketed by HIP events on its own stream
        m.set_profile(True)
        one_step()
        ps = m.stats()
        m.set_profile(False)
        if ps["score_ms"] > 0:
            achieved = ps["algo_bytes"] / (ps["score_ms"] * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                        "kernel": "k_score_afs (+k_score_afq for the first launches)" if args.af else "k_score_int",
                        "launches": ps["score_launches"],
                        "avg_launch_us": ps["score_ms"] * 1e3 / max(1, ps["score_launches"]),
                        "algo_bytes_per_launch": ps["algo_bytes"] / max(1, ps["score_launches"]),
                        "rank": rank}

    # PMC traffic cannot be read from inside the process: two counter passes in child processes (N = 1), else the
    # passes recorded under profiles/ for the default configurations (tools/summarize_profile.py)
    if roofline is not None and args.pmc_traffic != "off":
        # (the child holds a second copy of the matrix next to this process's: only when that is a small part of the HBM)
        fits_twice = args.n_var * n_total / 8 < 0.25 * device.nat.device_memory(dev_index)[1]
        live = live_pmc_traffic(args) if args.pmc_traffic == "live" and world == 1 and fits_twice else None
        if live is not None:
            roofline["traffic"], roofline["traffic_launches"], roofline["traffic_source"] = live
        elif args.n_var == 10_000_000 and n_total == 2504 and args.select < 0 and world == 1 \
                and not args.chunk_vars and (not args.af or args.af_dtype == "f32"):
            rec = os.path.join(ROOT, "profiles", "r01_cfg3_pmc_hbm.json" if args.af else "r01_cfg2_pmc_hbm.json")
            if os.path.exists(rec):
                with open(rec) as fh:
                    roofline["traffic"] = json.load(fh)["hbm_bytes_per_launch_mean"]
                roofline["traffic_source"] = os.path.relpath(rec, ROOT) + " (recorded rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"

    # N > 1: rank 0 re-runs the whole problem alone (untimed, own context) and compares the rows -- evidence
    # from this very run that the sharded exchange decides exactly like a single GPU
    sharded_check = None
    if (world > 1 or args.force_comm) and rank == 0 and not args.no_sharded_check:
        with device.DeviceMatrix(n_total, device=dev_index) as solo:
            v0 = 0
            while v0 < args.n_var:
                nv = min(chunk_vars, args.n_var - v0)
                c = solo.add_chunk(nv)
                solo.synth_fill(c, seed=args.seed, first_var_global=v0)
                if args.af:
                    _, af = device.synth_host(args.seed, nv, n_total, first_var_global=v0, want_cols=False)
                    solo.set_af(c, af if args.af_dtype == "f32" else af.astype(np.float64) / 3.0)
                v0 += nv
            s_idx, s_new, s_score = solo.run(k_sel)
        sharded_check = bool(len(s_idx) == len(idx) and (s_idx == idx).all() and (s_new == new).all())
        if not sharded_check:
            sys.stderr.write("bench.py: SHARDED RESULT DIFFERS FROM THE SINGLE-GPU RESULT\n")

    cpu = bitset = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, bitset = cpu_baseline(args, device)

    if id_path and rank == 0:
        try:
            os.remove(id_path)
        except OSError:
            pass
    if rank != 0:
        sync_max(0.0)      # nobody unmaps its columns while a peer may still be finishing
        if transport is not None:
            transport.close()
        m.close()
        return
    sync_max(0.0)
    value = iters / elapsed
    whole_loop_gbps = algo_bytes_step * world * args.steps / elapsed / 1e9   # shards are equal-sized to within one sample
    line = {
        "metric": "greedy iterations/sec + achieved HBM GB/s, 10M variants \u00d7 2.5k samples",   # BASELINE.json's string
        "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / max(1, args.steps) * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": (args.af_dtype + "+u64") if args.af else "u64", "data": "synthetic",
        "config": {"workload": f"synthetic {args.n_var} variants x {n_total} samples bit-matrix, select "
                               f"{'all' if args.select < 0 else k_sel}{(', --af ' + args.af_dtype) if args.af else ''}",
                   "n_var": args.n_var, "n_samp": n_total, "iterations_per_step": iters // max(1, args.steps),
                   "tot_captured": tot_captured, "chunks": st["n_chunks"], "seed": args.seed,
                   "sharding": f"sample axis over {world} GPU(s); per-iteration exchange: {exchange}" if world > 1 else "none",
                   "generator_s": round(t_gen, 3), "af_verified_parallel": st["af_fixed_point"] if args.af else None},
        "scoring": "decremental after the first passes (bytes = what this variant actually reads; NOT the brute-force "
                   "roofline metric)" if args.decremental else "brute force: every selectable column re-read every iteration",
        "decremental_iterations_per_step": st["decr_iterations"] if args.decremental else 0,
        "decremental_interleaved_copy_bytes": st["decr_interleaved_bytes"] if args.decremental else 0,
        "brute_force_equivalent_gbps": st["brute_force_bytes"] * world * args.steps / elapsed / 1e9,
        "hbm_gbps_whole_loop": whole_loop_gbps, "hbm_frac_whole_loop": whole_loop_gbps / (HBM_PEAK_GBPS * world),
        "device_loop_ms_per_step": loop_ms / max(1, args.steps),
        "sharded_rows_match_single_gpu": sharded_check,
        "roofline": roofline, "cpu_baseline": cpu, "cpu_bitset_baseline": bitset,
    }
    print(json.dumps(line))
    if transport is not None:
        transport.close()
    m.close()


if __name__ == "__main__":
    main()
