#!/usr/bin/env python3
"""bench.py -- greedy iterations/sec + achieved HBM GB/s of the selection loop on MI355X.

A *step* is one full `select all` run of the greedy loop (utmos/select.py:69-112) over the synthetic
10M-variant x 2,504-sample bit matrix (BASELINE.json configs[1]), matrix already resident in HBM.
`value` = greedy iterations per second over the K timed steps (all ranks, max time over ranks).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W          # one rank per GPU; samples sharded over the ranks

N = 1, default workload: after the headline the other single-GPU BASELINE configurations run a few timed steps each (cfg3
`--af` float32 and the same with the reference's in-memory float64 values, cfg1 chr22-sized, one rank's share of cfg4
for 20 iterations, cfg5's 156 GB for 10) and are attached under `also`, each with its own it/s, bytes and roofline fraction.

N > 1: the same 10M x 2,504 problem is sharded over the sample axis (strong scaling).  The line says how the shards
met every iteration (`exchange`, `rccl_ranks`, `p2p_replica_bytes`); the headline is the default exchange (device
mailboxes over hipIpc mappings) and the same steps are timed again with north_star's RCCL protocol (ncclAllGather
of the records + ncclBroadcast of the winner's column) under `also_exchange`.  No torch is imported: ranks find
each other through RANK/LOCAL_RANK/WORLD_SIZE and a rendezvous file.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
METRIC = "greedy iterations/sec + achieved HBM GB/s, 10M variants × 2.5k samples"   # BASELINE.json's string

# BASELINE.json `configs`, by name.  cfg2 is the default (the metric's configuration); cfg4 is meant for --gpus 8
# (12,500 columns = 78 GB per rank; `cfg4rank` is that one-rank share on one GPU) and cfg5 for one GPU (156 GB in ten
# chunks); both select a fixed number of samples instead of all of them.
WORKLOADS = {
    "cfg1": ("1.1M x 2,504 (chr22-sized), select all", dict(n_var=1_103_547, n_samp=2504, select=-1)),
    "cfg1af": ("1.1M x 2,504 (chr22-sized) with float32 AF weighting, select all", dict(n_var=1_103_547, n_samp=2504, select=-1, af=True)),
    "cfg1af64": ("1.1M x 2,504 (chr22-sized) with float64 AF values, select all",
                 dict(n_var=1_103_547, n_samp=2504, select=-1, af=True, af_dtype="f64")),
    "cfg2": ("10M x 2,504, select all", dict(n_var=10_000_000, n_samp=2504, select=-1)),
    "cfg3": ("10M x 2,504 with float32 AF weighting, select all", dict(n_var=10_000_000, n_samp=2504, select=-1, af=True)),
    "af64": ("10M x 2,504 with float64 AF values (the reference's in-memory --af), select all",
             dict(n_var=10_000_000, n_samp=2504, select=-1, af=True, af_dtype="f64")),
    "cfg4": ("50M x 100,000 over the ranks, first 20 iterations", dict(n_var=50_000_000, n_samp=100_000, select=20)),
    "cfg4rank": ("one rank's share of cfg4 on one GPU: 50M x 12,500, first 20 iterations",
                 dict(n_var=50_000_000, n_samp=12_500, select=20)),
    "cfg5": ("500M x 2,504 in chunks of 50M, first 10 iterations",
             dict(n_var=500_000_000, n_samp=2504, select=10, chunk_vars=50_000_000)),
}
ALSO = ("cfg3", "af64", "cfg1", "cfg1af", "cfg1af64", "cfg4rank", "cfg5")   # attached to the default single-GPU line, 1-5 timed steps each


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
    p.add_argument("--cpu-sample-vars", type=int, default=1_000_000, help="rows of the CPU baseline's sample when full N is not timed")
    p.add_argument("--cpu-budget-s", type=float, default=75.0,
                   help="CPU seconds the reference port may take; one iteration at full N is timed when it fits here and in host RAM")
    p.add_argument("--no-roofline-pass", action="store_true")
    p.add_argument("--no-calibration", action="store_true", help="N = 1: skip the streaming-read calibration leg")
    p.add_argument("--calibration-launches", type=int, default=20)
    p.add_argument("--no-also", action="store_true", help="N = 1: skip the other BASELINE configurations after the headline")
    p.add_argument("--af-estimate-scores", action="store_true",
                   help="AF: do not chain unambiguous winners (same rows; reported scores are estimates)")
    p.add_argument("--decremental", action="store_true",
                   help="SURVEY 8f-4 shortcut (exact, reads far fewer bytes): reported separately, never the default")
    p.add_argument("--no-sharded-check", action="store_true", help="N > 1: skip rank 0's single-GPU re-run and comparison")
    p.add_argument("--rccl-leg-timeout", type=float, default=240.0,
                   help="N > 1: seconds the additional RCCL measurements may take before the line is printed without them")
    p.add_argument("--exchange", choices=["auto", "mailboxes", "rccl", "rccl-allreduce", "both"], default="both",
                   help="N > 1: both = headline through the default exchange (mailboxes, else RCCL) and the same steps again "
                        "through RCCL (also_exchange); auto / mailboxes / rccl = that one only")
    p.add_argument("--decr-threshold", type=float, default=0.0, help="--decremental: newly-covered word fraction below which an iteration goes decremental (0 = library default)")
    p.add_argument("--pmc-traffic", choices=["live", "recorded", "off"], default="live",
                   help="roofline.traffic: live = two rocprofv3 --pmc child runs of one step of this workload (N = 1), "
                        "falling back to the passes recorded under profiles/; recorded = only those; off = null")
    p.add_argument("--workload", choices=sorted(WORKLOADS), default=None,
                   help="a BASELINE.json configuration by name (sets the shape flags): " +
                        "; ".join(f"{k}: {v[0]}" for k, v in sorted(WORKLOADS.items())))
    p.add_argument("--force-comm", action="store_true", help="initialise RCCL even with one rank (exercises the RCCL exchange)")
    p.add_argument("--force-mailboxes", action="store_true",
                   help="one rank: exchange through the device mailboxes anyway (the shard posts to and collects from itself)")
    args = p.parse_args()
    args.explicit_shape = bool(args.workload) or any(a.startswith(("--n-var", "--n-samp", "--select", "--af", "--chunk-vars",
                                                                   "--decremental", "--seed")) for a in sys.argv[1:])
    if args.workload:
        for key, value in WORKLOADS[args.workload][1].items():
            setattr(args, key, value)
    return args


# ------------------------------------------------------------------------------------------------ building blocks
def spec_of(args):
    return dict(n_var=args.n_var, n_samp=args.n_samp, select=args.select, af=args.af, af_dtype=args.af_dtype,
                chunk_vars=args.chunk_vars, seed=args.seed)


def build_matrix(device, spec, dev_index, first=0, n_local=None):
    """The synthetic matrix of `spec`, generated in HBM (DESIGN.md 'Synthetic input').  -> (matrix, seconds)."""
    import numpy as np
    n_total = spec["n_samp"]
    m = device.DeviceMatrix(n_total, device=dev_index, first_sample=first, n_local=n_total if n_local is None else n_local)
    chunk_vars = spec.get("chunk_vars") or spec["n_var"]
    t0 = time.perf_counter()
    v0 = 0
    while v0 < spec["n_var"]:
        nv = min(chunk_vars, spec["n_var"] - v0)
        c = m.add_chunk(nv)
        m.synth_fill(c, seed=spec.get("seed", 0), first_var_global=v0)
        if spec.get("af"):
            _, af = device.synth_host(spec.get("seed", 0), nv, n_total, first_var_global=v0, want_cols=False)
            # f64: full 53-bit mantissas, like the reference's ac/an quotients
            m.set_af(c, af if spec.get("af_dtype", "f32") == "f32" else af.astype(np.float64) / 3.0)
        v0 += nv
    return m, time.perf_counter() - t0


def select_count(spec):
    return spec["n_samp"] if spec["select"] < 0 else min(spec["select"], spec["n_samp"])


def timed_steps(m, k_sel, steps, warmup, sync_max):
    """W untimed + K timed steps, bracketed by barriers; the maximum time over the ranks.  A step = reset + run."""
    def one_step():
        m.reset()
        return m.run(k_sel)
    for _ in range(warmup):
        one_step()
    sync_max(0.0)                     # barrier (run() returns only after its stream has drained)
    t0 = time.perf_counter()
    iters, loop_ms, rows = 0, 0.0, None
    for _ in range(steps):
        rows = one_step()
        iters += len(rows[0])
        loop_ms += m.stats()["loop_ms"]
    elapsed = sync_max(time.perf_counter() - t0)
    return dict(iters=iters, elapsed=elapsed, loop_ms=loop_ms, rows=rows, stats=m.stats())


def roofline_pass(m, k_sel, af, rank=0):
    """The same step once more with every scoring dispatch stamped by its own HIP events (on the context's stream):
    algorithmic bytes of the step / summed scoring-kernel time."""
    m.set_profile(True)
    m.reset()
    m.run(k_sel)
    ps = m.stats()
    m.set_profile(False)
    if ps["score_ms"] <= 0:
        return None
    achieved = ps["algo_bytes"] / (ps["score_ms"] * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None,
            "kernel": ("k_loop_int, AF form (persistent: up to 256 delta iterations per launch) behind the first k_score_afq / k_score_afs passes"
                       if af and ps["persist_iterations"] else
                       "k_score_afs (+ k_score_afq for the first launches)" if af else
                       "k_loop_int (persistent: up to 256 iterations per launch, pick inside)" if ps["persist_iterations"] else
                       "k_score_int (pick fused in on one GPU)"),
            "launches": ps["score_launches"], "iterations_in_persistent_launches": ps["persist_iterations"],
            "avg_launch_us": ps["score_ms"] * 1e3 / max(1, ps["score_launches"]),
            "algo_bytes_per_launch": ps["algo_bytes"] / max(1, ps["score_launches"]), "rank": rank}


def live_pmc_traffic(spec, extra):
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
    work = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline-pass", "--no-also", "--no-calibration", "--pmc-traffic", "off",
            "--n-var", str(spec["n_var"]), "--n-samp", str(spec["n_samp"]), "--select", str(spec["select"]),
            "--seed", str(spec["seed"]), "--chunk-vars", str(spec["chunk_vars"])] + extra
    if spec["af"]:
        work += ["--af", "--af-dtype", spec["af_dtype"]]
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
                                       if ("k_score_" in r["Kernel_Name"] or "k_loop_int" in r["Kernel_Name"]) and r["Counter_Name"] == counter]
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


def mem_available_gb():
    try:
        with open("/proc/meminfo") as fh:
            for line in fh:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def cpu_baseline(args, device_mod, gpu_winners):
    """The reference's algorithm on the host cores (rank 0, N = 1): oracle.score_rowloop = the per-row numpy loop of
    utmos/select.py:33-48, one thread.  BASELINE.md 4 asks for full N when host RAM allows the unpacked bool matrix; one
    full-N iteration costs 40-95 s depending on the host (calibrated on a 200k-row slice first), so it is timed only
    when that fits --cpu-budget-s too -- otherwise the first
    `cpu_sample_vars` variants of the same matrix: the first 3 iterations (nothing captured yet: the most expensive
    ones) and one iteration from the state the GPU run had reached after S/2 selections (captured rows are skipped
    there); rates scaled to the full variant count (time per iteration is linear in rows).
    Second line: the packed C oracle with OpenMP."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_util as ou
    ram_gb = mem_available_gb()
    full_bytes_gb = args.n_var * args.n_samp / 1e9 * 2.1         # bool matrix + the unpacked bits it is transposed from
    # per-row cost of the reference's loop on THIS host: 9-10 us measured with the reference itself in the build
    # container, ~4 us on the GPU boxes' hosts -- so calibrate on a slice before deciding
    n_cal = min(args.n_var, 200_000)
    with device_mod.DeviceMatrix(args.n_samp, device=0) as m:
        c = m.add_chunk(n_cal)
        m.synth_fill(c, seed=args.seed)
        cal = m.download_columns(c)
    cal = np.ascontiguousarray(np.unpackbits(cal.view(np.uint8), axis=1, bitorder="little")[:, :n_cal].T).astype(bool)
    t0 = time.perf_counter()
    ou.npo.score_rowloop(cal, np.ones(args.n_samp, np.uint8))
    per_row = (time.perf_counter() - t0) / n_cal
    del cal
    full_est_s = args.n_var * per_row
    full_n = ram_gb > full_bytes_gb + 8 and full_est_s * 1.3 < args.cpu_budget_s
    n_var = args.n_var if full_n else min(args.cpu_sample_vars, args.n_var)
    why = ("full N" if full_n else
           f"sample: full N needs {full_bytes_gb:.0f} GB of host RAM ({ram_gb:.0f} GB available) and ~{full_est_s:.0f} s per "
           f"iteration on this host ({per_row * 1e6:.1f} us/row calibrated; budget {args.cpu_budget_s:.0f} s)")
    with device_mod.DeviceMatrix(args.n_samp, device=0) as m:
        c = m.add_chunk(n_var)
        m.synth_fill(c, seed=args.seed)
        cols = m.download_columns(c)
    bits = np.unpackbits(cols.view(np.uint8), axis=1, bitorder="little")[:, :n_var]
    dense = np.ascontiguousarray(bits.T).astype(bool)
    del bits
    scale = n_var / args.n_var
    state = np.ones(args.n_samp, np.uint8)
    iters = 1 if full_n else 3
    t0 = time.perf_counter()
    for _ in range(iters):
        best, _new = ou.npo.score_rowloop(dense, state)
        state[best] = 0
    dt = time.perf_counter() - t0
    port = {"value": iters / dt * scale, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"first {iters} greedy iteration(s) on {'all' if full_n else 'the first'} {n_var} of {args.n_var} synthetic "
                      f"variants x {args.n_samp} samples (numpy row loop, {dt:.1f} s); rate scaled by {scale:.4g} to the full variant count",
            "rows_timed": n_var, "full_n": bool(full_n), "why": why, "us_per_row_calibrated": round(per_row * 1e6, 2),
            "host_cpus": os.cpu_count(), "host_mem_available_gb": round(ram_gb, 1)}
    # a later iteration skips the rows already captured, so it is cheaper than the first: time one from the state the
    # GPU run had reached after S/2 selections (at full N only when the budget still has room for it)
    if gpu_winners is not None and len(gpu_winners) >= 2 and (not full_n or dt * 1.6 < args.cpu_budget_s):
        half = len(gpu_winners) // 2
        mid = np.ones(args.n_samp, np.uint8)
        mid[np.asarray(gpu_winners[:half])] = 0          # the GPU run's first S/2 winners are "used": their variants are covered
        t0 = time.perf_counter()
        ou.npo.score_rowloop(dense, mid)
        dt_mid = time.perf_counter() - t0
        port["mid_run"] = {"value": 1.0 / dt_mid * scale, "unit": "iterations/s",
                           "sample": f"one iteration after {half} selections (state taken from the GPU run), same rows, {dt_mid:.1f} s"}
    threads = min(os.cpu_count() or 1, 16)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    t0 = time.perf_counter()
    idx, _, _ = ou.c_greedy(cols, n_var, np.ones(args.n_samp, np.uint8), k_max=40, omp=True)
    dt2 = time.perf_counter() - t0
    bitset = {"value": len(idx) / dt2 * scale, "unit": "iterations/s", "cores": threads, "kind": "port",
              "sample": f"first {len(idx)} iterations, packed C bitset oracle with OpenMP, same rows, scaled the same way"}
    return port, bitset


def summarize(spec, label, world, steps, res, roofline, t_gen):
    """Per-configuration record (the `also` entries and the core of the headline)."""
    st = res["stats"]
    gbps = st["algo_bytes"] * world * steps / res["elapsed"] / 1e9        # shards are equal-sized to within one sample
    return {"workload": label, "value": res["iters"] / res["elapsed"], "unit": "iterations/s",
            "ms_per_step": res["elapsed"] / max(1, steps) * 1e3, "steps": steps,
            "iterations_per_step": res["iters"] // max(1, steps), "tot_captured": st["tot_captured"], "chunks": st["n_chunks"],
            "algo_bytes_per_step": st["algo_bytes"] * world, "hbm_gbps_whole_loop": gbps,
            "hbm_frac_whole_loop": gbps / (HBM_PEAK_GBPS * world), "device_loop_ms_per_step": res["loop_ms"] / max(1, steps),
            **({"whole_loop_frac_of_stream": gbps / roofline["stream_calibration_gbps"]} if roofline and roofline.get("stream_calibration_gbps") else {}),
            "generator_s": round(t_gen, 3),
            **({k: st[k] for k in ("af_chained_iterations", "af_deferred_rows")} if spec["af"] else {}),  # (of the last step)
            "roofline": None if roofline is None else {k: roofline.get(k) for k in ("frac", "achieved", "kernel", "launches", "avg_launch_us",
                                                                                   "algo_bytes_per_launch", "iterations_in_persistent_launches",
                                                                                   "stream_calibration_gbps", "frac_of_stream")}}


def workload_label(spec):
    k = select_count(spec)
    return (f"synthetic {spec['n_var']} variants x {spec['n_samp']} samples bit-matrix, select "
            f"{'all' if spec['select'] < 0 else k}{(', --af ' + spec['af_dtype']) if spec['af'] else ''}")


def headline_line(args, world, n_total, exchange, exchange_note, final_stats, st, res, head, roofline):
    """The JSON line's fields that the headline measurement alone decides (the legs behind it fill in the rest)."""
    from utmos_amd import _native
    env_overrides = _native.env_overrides()
    return {
        "metric": METRIC, "value": head["value"], "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": (args.af_dtype + "+u64") if args.af else "u64", "data": "synthetic",
        "config": {"workload": head["workload"], "n_var": args.n_var, "n_samp": n_total,
                   "iterations_per_step": head["iterations_per_step"], "tot_captured": head["tot_captured"], "chunks": head["chunks"],
                   "seed": args.seed, "sharding": f"sample axis over {world} GPU(s)" if world > 1 else "none",
                   "generator_s": head["generator_s"], "af_verified_parallel": st["af_fixed_point"] if args.af else None,
                   **{k: head[k] for k in ("af_chained_iterations", "af_deferred_rows") if k in head}},
        "exchange": exchange, "exchange_note": exchange_note, "rccl_ranks": final_stats["rccl_ranks"] if exchange.startswith("rccl") else None,
        "p2p_replica_bytes": final_stats["p2p_replica_bytes"] if world > 1 else None,
        "also_exchange": None,
        "scoring": "decremental after the first passes (bytes = what this variant actually reads; NOT the brute-force "
                   "roofline metric)" if args.decremental else "brute force: every selectable column re-read every iteration",
        "decremental_iterations_per_step": st["decr_iterations"] if args.decremental else 0,
        "decremental_interleaved_copy_bytes": st["decr_interleaved_bytes"] if args.decremental else 0,
        "brute_force_equivalent_gbps": st["brute_force_bytes"] * world * args.steps / res["elapsed"] / 1e9,
        "hbm_gbps_whole_loop": head["hbm_gbps_whole_loop"], "hbm_frac_whole_loop": head["hbm_frac_whole_loop"],
        "whole_loop_frac_of_stream": head.get("whole_loop_frac_of_stream"),
        "device_loop_ms_per_step": head["device_loop_ms_per_step"],
        "sharded_rows_match_single_gpu": None,
        # UTM_* knobs set in this process's environment (launch shapes / thresholds / test hooks, never results): a
        # value left behind by an experiment must not go unnoticed in a recorded line
        "env_overrides": env_overrides,
        "roofline": roofline, "cpu_baseline": None, "cpu_bitset_baseline": None, "also": None,
    }


# ------------------------------------------------------------------------------------------------ main
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

    spec = spec_of(args)
    n_total = args.n_samp
    first = rank * n_total // world
    n_local = (rank + 1) * n_total // world - first
    dev_index = local_rank % device.nat.device_count()   # one GPU per rank on a full node; ranks share devices only on smaller test boxes
    m, t_gen = build_matrix(device, spec, dev_index, first, n_local)
    k_sel = select_count(spec)
    if args.decremental:
        m.set_decremental(True, args.decr_threshold)
    if args.af_estimate_scores:
        m.set_af_exact_scores(False)

    transport = None
    exchange = "none"
    uid = None
    if world > 1:
        # start-up over a local TCP socket (port, nonce and ncclUniqueId published in the launch's rendezvous file)
        from utmos_amd.sharded import bootstrap, connect_shards
        transport, uid = bootstrap(rank, world, device.DeviceMatrix.comm_unique_id)
        try:
            exchange = connect_shards(m, transport, uid, "auto" if args.exchange == "both" else args.exchange)
        except RuntimeError as err:
            raise SystemExit(f"bench.py rank {rank}: {err}")
    elif args.force_mailboxes:
        # the ONE shard exports to and imports from itself, passes the self-test and posts its record to its own mailbox every
        # iteration: the device-side exchange's per-iteration cost without a second GPU
        m.p2p_import(0, [m.p2p_export()])
        if not m.p2p_selftest():
            raise SystemExit("bench.py: the mailbox self-test failed on this device")
        m.p2p_use_mailboxes("single")
        exchange = "mailboxes"
    elif args.force_comm:
        m.comm_init(0, 1, device.DeviceMatrix.comm_unique_id())
        exchange = "rccl"
        if args.exchange == "rccl-allreduce":
            m.comm_column_by_allreduce(True)
            exchange = "rccl-allreduce"

    def sync_max(value):
        """Barrier + maximum over the ranks (host side: the ranks' loops are already drained)."""
        if transport is None:
            return value
        return max(r[0] for r in transport.allgather((float(value), 0, 0)))

    # N = 1: what this box streams today -- plain read-only passes over the resident matrix with the scoring kernel's access
    # shape, timed in this process right before the headline (boxes differ by a few percent; the spec peak does not say)
    calibration_gbps = None
    if world == 1 and not args.no_calibration:
        calibration_gbps = m.stream_calibration(max(20, args.calibration_launches))

    exchange_note = None
    try:
        res = timed_steps(m, k_sel, args.steps, args.warmup, sync_max)
        failed = None
    except device.nat.NativeError as exc:
        if transport is None:
            raise
        res, failed = None, exc
    if transport is not None and not transport.agree(failed is None):
        # a rank lost a record in the mailboxes (the wait is bounded: every rank comes back within seconds): the same
        # steps through RCCL instead, and the line says so
        if exchange != "mailboxes":
            raise SystemExit(f"bench.py rank {rank}: the {exchange} exchange failed ({failed})")
        exchange_note = f"the device mailboxes passed their self-test but failed in the loop ({failed}); measured through RCCL (root-free form) instead"
        m.p2p_use_mailboxes(False)
        err = None
        try:
            m.comm_init(rank, world, uid)
        except device.nat.NativeError as exc:
            err = exc
        if not transport.agree(err is None):
            raise SystemExit(f"bench.py rank {rank}: no working exchange (mailboxes: {failed}; RCCL: {err})")
        m.comm_column_by_allreduce(True)
        exchange = "rccl-allreduce"
        args.exchange = "rccl-allreduce"
        res = timed_steps(m, k_sel, args.steps, args.warmup, sync_max)
    st = res["stats"]
    idx, new, _ = res["rows"]

    roofline = None
    if not args.no_roofline_pass and not args.decremental:   # the roofline object describes the brute-force kernel only
        roofline = roofline_pass(m, k_sel, args.af, rank)
        if roofline is not None and calibration_gbps:
            roofline["stream_calibration_gbps"] = calibration_gbps
            roofline["frac_of_stream"] = roofline["achieved"] / calibration_gbps
            roofline["stream_calibration"] = ("k_stream_read: read-only, 16 B per lane non-temporal, 1 KiB per wave instruction, 8 in flight, "
                                              f"{max(20, args.calibration_launches)} passes over the resident matrix in this process before the headline")

    # N > 1: the same steps again through north_star's RCCL protocol, measured beside the default exchange.  These legs
    # bring up a communicator that the headline did not need: should that, or a collective, hang on some node, every
    # rank leaves after --rccl-leg-timeout seconds and rank 0 still prints the headline it has already measured.
    also_exchange = None
    early = {"line": None}
    watchdog = None
    real_stdout = os.dup(1)
    if world > 1 and args.exchange == "both" and exchange == "mailboxes":
        import threading

        def give_up():
            if rank == 0 and early["line"] is not None:
                early["line"]["also_exchange"] = {"error": f"the RCCL legs did not finish within {args.rccl_leg_timeout:.0f} s (communicator set-up "
                                                           f"or a collective hung); the headline was measured through {exchange} before them"}
                # (straight to the process's own stdout: while a communicator is being set up, fd 1 is parked on stderr
                # to keep RCCL's banner out of the line -- device.comm_init)
                os.write(real_stdout, (json.dumps(early["line"]) + "\n").encode())
            # measured, but the RCCL legs hung: the line is out, the exit status says so (4); no headline at all: 3
            os._exit(4 if early["line"] is not None or rank != 0 else 3)

        if rank == 0:
            early["line"] = headline_line(args, world, n_total, exchange, exchange_note, m.stats(), st, res,
                                          summarize(spec, workload_label(spec), world, args.steps, res, roofline, t_gen), roofline)
        watchdog = threading.Timer(args.rccl_leg_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        m.p2p_use_mailboxes(False)
        err = None
        try:
            m.comm_init(rank, world, uid)
        except device.nat.NativeError as exc:
            err = exc
        if transport.agree(err is None):
            r2 = timed_steps(m, k_sel, args.steps, min(args.warmup, 1), sync_max)
            s2 = r2["stats"]
            same = bool(len(r2["rows"][0]) == len(idx) and (r2["rows"][0] == idx).all() and (r2["rows"][1] == new).all())
            also_exchange = {"rccl": {"value": r2["iters"] / r2["elapsed"], "unit": "iterations/s",
                                      "ms_per_step": r2["elapsed"] / max(1, args.steps) * 1e3, "exchange": m.exchange(),
                                      "rccl_ranks": s2["rccl_ranks"], "rows_match_default_exchange": same,
                                      "protocol": "per iteration: ncclAllGather of 64-byte records, ncclBroadcast of the winner's column from its owner "
                                                  "(one host sync per iteration: the root is data dependent)"}}
            m.comm_column_by_allreduce(True)
            r3 = timed_steps(m, k_sel, args.steps, min(args.warmup, 1), sync_max)
            same3 = bool(len(r3["rows"][0]) == len(idx) and (r3["rows"][0] == idx).all() and (r3["rows"][1] == new).all())
            also_exchange["rccl_allreduce"] = {
                "value": r3["iters"] / r3["elapsed"], "unit": "iterations/s", "ms_per_step": r3["elapsed"] / max(1, args.steps) * 1e3,
                "exchange": m.exchange(), "rccl_ranks": r3["stats"]["rccl_ranks"], "rows_match_default_exchange": same3,
                "protocol": "per iteration: ncclAllGather of 64-byte records, ncclAllReduce(sum) of owner's-column-else-zeros (root-free, no host sync)"}
            m.comm_column_by_allreduce(False)
        else:
            also_exchange = {"rccl": {"error": f"RCCL communicator unavailable on some rank ({err})"}}
        m.p2p_use_mailboxes(True)             # back to the headline's exchange for the check below
        sync_max(0.0)
        watchdog.cancel()
    os.close(real_stdout)

    # PMC traffic cannot be read from inside the process: two counter passes in child processes (N = 1), else the
    # passes recorded under profiles/ for the default configurations (tools/summarize_profile.py)
    if roofline is not None and args.pmc_traffic != "off":
        # (the child holds a second copy of the matrix next to this process's: only when that is a small part of the HBM)
        fits_twice = args.n_var * n_total / 8 < 0.25 * device.nat.device_memory(dev_index)[1]
        extra = ["--af-estimate-scores"] if args.af_estimate_scores else []
        live = live_pmc_traffic(spec, extra) if args.pmc_traffic == "live" and world == 1 and fits_twice else None
        if live is not None:
            roofline["traffic"], roofline["traffic_launches"], roofline["traffic_source"] = live
        elif args.n_var == 10_000_000 and n_total == 2504 and args.select < 0 and world == 1 \
                and not args.chunk_vars and (not args.af or args.af_dtype == "f32"):
            rec = os.path.join(ROOT, "profiles", "r02_cfg3_pmc_hbm.json" if args.af else "r02_cfg2_pmc_hbm.json")
            if os.path.exists(rec):
                with open(rec) as fh:
                    roofline["traffic"] = json.load(fh)["hbm_bytes_per_launch_mean"]
                roofline["traffic_source"] = os.path.relpath(rec, ROOT) + " (recorded rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"

    # N > 1: rank 0 re-runs the whole problem alone (untimed, own context) and compares the rows -- evidence
    # from this very run that the sharded exchange decides exactly like a single GPU
    sharded_check = None
    if (world > 1 or args.force_comm or args.force_mailboxes) and rank == 0 and not args.no_sharded_check:
        solo, _ = build_matrix(device, spec, dev_index)
        with solo:
            s_idx, s_new, _ = solo.run(k_sel)
        sharded_check = bool(len(s_idx) == len(idx) and (s_idx == idx).all() and (s_new == new).all())
        if not sharded_check:
            sys.stderr.write("bench.py: SHARDED RESULT DIFFERS FROM THE SINGLE-GPU RESULT\n")

    head = summarize(spec, workload_label(spec), world, args.steps, res, roofline, t_gen)
    final_stats = m.stats()
    sync_max(0.0)          # nobody unmaps its columns while a peer may still be finishing
    if transport is not None:
        transport.close()
    m.close()
    if rank != 0:
        return

    # N = 1, default workload: the other single-GPU BASELINE configurations, a few steps each
    also = None
    if world == 1 and not args.explicit_shape and not args.no_also and not args.force_comm and not args.force_mailboxes:
        also = {}
        for name in ALSO:
            s2 = dict(af=False, af_dtype="f32", chunk_vars=0, seed=args.seed)
            s2.update(WORKLOADS[name][1])
            try:
                m2, t2 = build_matrix(device, s2, dev_index)
                with m2:
                    k2 = select_count(s2)
                    cal2 = m2.stream_calibration(20 if s2["n_var"] * s2["n_samp"] < 4e11 else 3)
                    # timed steps per leg: 5 where a step is under 0.1 s, 2 at 10M x 2,504, 1 for the 78 / 156 GB shapes
                    n2 = 5 if name in ("cfg1", "cfg1af", "cfg1af64") else 2 if name in ("cfg3", "af64") else 1
                    r2 = timed_steps(m2, k2, n2, 1 if n2 > 1 else 0, lambda v: v)
                    roof2 = roofline_pass(m2, k2, s2["af"])
                    if roof2 is not None:
                        roof2["stream_calibration_gbps"] = cal2
                        roof2["frac_of_stream"] = roof2["achieved"] / cal2
                also[name] = summarize(s2, WORKLOADS[name][0], 1, n2, r2, roof2, t2)
            except device.nat.NativeError as exc:      # e.g. a GPU with less HBM than the 156 GB of cfg5
                also[name] = {"workload": WORKLOADS[name][0], "error": str(exc)}

    cpu = bitset = None
    if world == 1 and not args.no_cpu_baseline:
        cpu, bitset = cpu_baseline(args, device, idx)

    line = headline_line(args, world, n_total, exchange, exchange_note, final_stats, st, res, head, roofline)
    line.update({"also_exchange": also_exchange, "sharded_rows_match_single_gpu": sharded_check, "cpu_baseline": cpu,
                 "cpu_bitset_baseline": bitset, "also": also})
    print(json.dumps(line))


if __name__ == "__main__":
    main()
