#!/usr/bin/env python3
"""bench.py -- headline benchmark of the DP hot path on MI355X.

Metric (BASELINE.json): DP cells/sec (GCUPS) on 16 kbp x 16 kbp pairs, 1/2/4/8-GPU batch
scaling.  Workload at every N: each GPU aligns its share of config 4 (1024 synthetic
circular 16 kbp pairs sharded over 8 GPUs = 128 pairs per GPU, weak scaling); a "step" is
one pass of the hot path -- matrix fill + direction traceback -- over that batch, inputs
(packed sequences / profile tables) already resident in HBM.  One process per GPU
(torch.distributed.run), no data-path collective; barriers and max-over-ranks timing only.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs P] [--len L]

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ALG_BYTES_PER_CELL = 0.25             # 2-bit direction per cell, the figure for roofline.achieved
# The benchmarked path is the bit-parallel pair kernel nw_fill_bits (csadp_bits.hip): a lane
# advances one row of 32 columns per step on three bit planes.  Default = checkpoint mode
# (CSADP_BITS_CKPT=1): the fill writes lane-state checkpoints (0.017 B/cell) instead of the
# direction planes and the traceback (nw_traceback_replay) re-derives the directions of the blocks
# on the path, so the implementation moves ~7 % of the algorithmic 0.25 B/cell and
# roofline.achieved (algorithmic bytes / time, as SURVEY 8(d) defines it) can exceed what the HBM
# could stream.  CSADP_BITS_CKPT=0 writes the direction planes (the kernel is then bound by HBM
# writes: 5.2 of the 5.9 TB/s a plain store kernel reaches on this chip, tools/hbm_write_probe.hip).
# Second roofline, VALU issue: ISA count of the compiled steady-state step = 20 v_bitop3 (2.6 issue
# cycles per wave64 and SIMD when measured alone), 10 DPP / three-operand instructions (4.3) and 2
# two-operand ones (2.1) per 32 cells -- tools/valu_microbench.hip, profiles/r01_valu_microbench.txt.
# CSADP_BITS=0 falls back to the packed-16 kernel (nw_fill_tiles_pk, 4 VALU per cell) and
# CSADP_BITS=0 CSADP_PK16=0 to the 32-bit kernel (6 per cell).
BITS = os.environ.get("CSADP_BITS", "1") != "0"
CKPT = os.environ.get("CSADP_BITS_CKPT", "1") != "0"
PK16 = os.environ.get("CSADP_PK16", "1") != "0"
if BITS:
    mix = {"v_bitop3": (20 if CKPT else 21, 2.6), "dpp_or_three_operand": (10, 4.3), "two_operand": (2 if CKPT else 3, 2.1)}
    VALU_OPS_PER_CELL = round(sum(n for n, _ in mix.values()) / 32, 3)
    VALU_ISSUE_CYCLES_PER_CELL_WAVE = round(sum(n * c for n, c in mix.values()) / 32, 3)
    FILL_KERNEL = "nw_fill_bits"
    DTYPE = "u32 bit planes"
else:
    VALU_OPS_PER_CELL = 4 if PK16 else 6
    VALU_ISSUE_CYCLES_PER_CELL_WAVE = 14 if PK16 else 18
    FILL_KERNEL = "nw_fill_tiles_pk" if PK16 else "nw_fill_tiles"
    DTYPE = "int16" if PK16 else "int32"
VALU_PEAK_CUPS = 256 * 4 * 64 / VALU_ISSUE_CYCLES_PER_CELL_WAVE * 2.4e9


def pmc_traffic_per_launch():
    """HBM bytes per fill launch from the committed PMC passes (rocprofv3 --pmc WRITE_SIZE /
    FETCH_SIZE in separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")) as f:
            k = json.load(f)[FILL_KERNEL]
        return int(k["hbm_write_bytes_per_launch"] + k["hbm_read_bytes_per_launch_x2_corrected"])
    except Exception:
        return None


def cpu_baseline(tasks, gpu_results, seconds_budget=25.0):
    """Time the CPU path on a bounded sample of the SAME workload on this host, 1 core:
    the compiled reference (oracle/_ref, kind 'reference') when its prebuilt library
    travelled with the repo, else the oracle port.  Also cross-checks the GPU strings."""
    from helpers import have_ref, oracle_progressive, ref_progressive
    try:  # keep freed matrix pages in the heap between calls (glibc would unmap and re-fault 1.4 GB per pair)
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-1, 1 << 30)   # M_TRIM_THRESHOLD
        libc.mallopt(-3, 1 << 25)   # M_MMAP_THRESHOLD (32 MiB is the glibc maximum)
    except Exception:
        pass
    kind = "reference" if have_ref() else "port"
    run = ref_progressive if kind == "reference" else (lambda t, r: oracle_progressive(t, r))
    cells = 0
    secs = 0.0
    n = 0
    mismatches = 0
    run(tasks[0][0], tasks[0][1])          # untimed warm-up: first touch of the 1.4 GB matrices
    for t, g in zip(tasks, gpu_results):
        t0 = time.perf_counter()
        out = run(t[0], t[1])
        dt = time.perf_counter() - t0
        if kind == "reference":
            dt = out[2]                    # wall time of ProgressiveDP alone, measured inside the shim
        secs += dt
        cells += len(t[0][0]) * len(t[0][1])
        n += 1
        if out[1] != g["aligned"]:
            mismatches += 1
        if secs > seconds_budget:
            break
    out = {"value": round(cells / secs / 1e9, 4), "unit": "GCUPS", "cores": 1, "kind": kind,
           "sample": "%d of the step's 16 kbp pairs, whole ProgressiveDP (fill+traceback), %.1f s, "
                     "GPU strings %s" % (n, secs, "identical" if mismatches == 0 else "MISMATCH x%d" % mismatches)}
    return out


def cpu_many_cores(tasks, per_proc=2):
    """The same CPU code in P independent processes (the reference keeps global state, so no
    threads), `per_proc` pairs each: what a host-only deployment of the reference would reach on a
    share of this node's cores.  Reported beside the single-core figure, wall clock of the slowest
    process.  Bounded: ~1.5 s per pair and process.  Runs BEFORE this process touches the GPU, so
    the forked children never hold a device context."""
    from helpers import have_ref, oracle_progressive, ref_progressive
    run = ref_progressive if have_ref() else (lambda t, r: oracle_progressive(t, r))
    procs = max(1, min(32, (os.cpu_count() or 2) // 2, len(tasks) // per_proc))
    pids = []
    t0 = time.perf_counter()
    for i in range(procs):
        pid = os.fork()
        if pid == 0:
            try:
                for t in tasks[i * per_proc:(i + 1) * per_proc]:
                    run(t[0], t[1])
                os._exit(0)
            except BaseException:
                os._exit(1)
        pids.append(pid)
    ok = True
    for pid in pids:
        _, st = os.waitpid(pid, 0)
        ok = ok and os.WIFEXITED(st) and os.WEXITSTATUS(st) == 0
    wall = time.perf_counter() - t0
    cells = sum(len(t[0][0]) * len(t[0][1]) for t in tasks[:procs * per_proc])
    return {"value": round(cells / wall / 1e9, 3) if ok else None, "unit": "GCUPS", "cores": procs,
            "sample": "%d processes x %d pairs, %.1f s wall" % (procs, per_proc, wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=128, help="pairs per GPU (config 4: 1024 / 8)")
    ap.add_argument("--len", type=int, default=16384, dest="length")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for barriers/reductions; 'gloo' + --share-device rehearses the "
                         "N>1 flow on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="all ranks use HIP device 0 (rehearsal only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # invoked plainly: start one child per GPU before anything here has touched a device
        from csa_amd.dist import spawn_ranks
        raise SystemExit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    many_cores = None
    if args.gpus == 1 and not args.no_cpu_baseline:
        from csa_amd.synth import config4_tasks as _tasks      # numpy only: no device is initialised here
        many_cores = cpu_many_cores(_tasks(0, min(args.pairs, 64), args.length))

    import torch
    import csa_amd
    from csa_amd import dist as cdist
    from csa_amd.synth import config4_tasks

    rank, local_rank, world = cdist.env_world()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    dev = 0 if args.share_device else local_rank
    if args.share_device:
        os.environ["CSADP_SHARE_DEVICE"] = "1"
    if args.backend == "nccl":
        torch.cuda.set_device(dev)
    csa_amd.init(device=dev)
    group = cdist.Group(backend=args.backend, device="cuda:%d" % dev)

    # weak scaling: rank r owns global pairs [r*P, (r+1)*P) of the synthetic batch
    tasks = config4_tasks(rank * args.pairs, args.pairs, args.length)
    t_c0 = time.perf_counter()
    batch = csa_amd.PairBatch(tasks)            # validate, pack, upload: inputs now resident in HBM
    create_s = time.perf_counter() - t_c0

    def sync():
        batch.sync()
        torch.cuda.synchronize()

    elapsed = cdist.timed_steps(group, batch.run, sync, args.steps, args.warmup)
    tm_pipe = batch.timing()                     # HIP events of the launch that held the LAST timed pass
    # one launch alone (nothing else in flight): the bit-parallel path merges `launch_passes`
    # consecutive passes into a launch, so request that many
    for _ in range(max(tm_pipe["launch_passes"], 1)):
        batch.run()
    sync()
    tm = batch.timing()
    cells_step = group.sum(tm["cells"])
    t_f0 = time.perf_counter()
    results = batch.fetch()                      # ops D2H + aligned strings built on the host
    fetch_s = time.perf_counter() - t_f0
    ok = all(r["status"] == 0 for r in results)

    value = cells_step * args.steps / elapsed / 1e9
    line = None
    if rank == 0:
        from helpers import degap, rotated, sp_score
        # properties on this rank's first pairs: strings re-spell the inputs, SP == DP score
        for t, r in list(zip(tasks, results))[:4]:
            ok = ok and degap(r["aligned"][0]) == rotated(t[0][0], t[1][0]) and degap(r["aligned"][1]) == rotated(t[0][1], t[1][1])
            ok = ok and sp_score(r["aligned"]) == r["score"]
        launches = max(tm["fill_launches"], 1)
        lp = max(tm["launch_passes"], 1)                        # passes per launch (1 unless bit-parallel)
        impl_bytes = tm["dir_bytes"] + tm["border_bytes"]      # what the kernels write per pass (directions or checkpoints, + tile borders)
        alg_bytes = ALG_BYTES_PER_CELL * tm["cells"] + (tm["border_bytes"] if not tm["bit_parallel"] else 0)
        # the fill kernel is in flight during the whole timed region (launches rotate over three
        # streams, the traceback of one hides under the next fill), so its sustained rate is:
        # work of all timed passes / wall time of the timed region
        rank_cells = tm["cells"]
        eff_bytes_s = alg_bytes * args.steps / elapsed
        eff_cups = rank_cells * args.steps / elapsed
        line = {
            "metric": "DP cells/sec (GCUPS) on 16 kbp x 16 kbp pairs, fill + traceback, whole job",
            "value": round(value, 3), "unit": "GCUPS", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE,
            "data": "synthetic", "verified": bool(ok),
            "per_gpu_gcups": round(value / args.gpus, 3),
            "config": {"workload": "config 4 share: %d synthetic circular %d bp pairs per GPU "
                                   "(global pairs rank*%d..), linear-gap NW fill + traceback, "
                                   "bit-exact vs reference" % (args.pairs, args.length, args.pairs),
                       "pairs_per_gpu": args.pairs, "seq_len": args.length,
                       "kernel": FILL_KERNEL,
                       "passes_per_launch": lp,
                       "parallelism": "tasks sharded over %d GPU(s), no collective" % args.gpus},
            "kernel_ms": {"fill_launch_pipelined": round(tm_pipe["fill_ms"], 3),
                          "traceback_launch_pipelined": round(tm_pipe["traceback_ms"], 3),
                          "fill_launch_alone": round(tm["fill_ms"], 3), "traceback_launch_alone": round(tm["traceback_ms"], 3),
                          "passes_per_launch": lp,
                          "fill_launches_per_pass": tm["fill_launches"] / lp, "fill_tiles": tm["fill_tiles"],
                          "fill_alone_gcups": round(lp * tm["cells"] / tm["fill_ms"] / 1e6, 2)},
            "host_boundary_ms": {"create_pack_upload": round(create_s * 1e3, 2), "fetch_download_strings": round(fetch_s * 1e3, 2),
                                 "pcie_inclusive_gcups": round(rank_cells / (create_s + fetch_s + tm["total_ms"] / lp / 1e3) / 1e9, 1),
                                 "note": "not part of value: one batch from host buffers to host strings, unpipelined"},
            "roofline": {"bound": "hbm", "kernel": FILL_KERNEL,
                         "achieved": round(eff_bytes_s / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(eff_bytes_s / 1e9 / HBM_PEAK_GBS, 6),
                         "traffic": pmc_traffic_per_launch(),
                         "bytes_per_launch": round(lp * alg_bytes / launches),
                         "avg_launch_us": round(tm["fill_ms"] * 1e3 / launches, 2),
                         "per_launch_achieved": round(lp * alg_bytes / (tm["fill_ms"] * 1e-3) / 1e9, 1),
                         "launch_us_pipelined": round(tm_pipe["fill_ms"] * 1e3 / launches, 2),
                         "implementation_bytes_per_launch": round(lp * impl_bytes / launches),
                         "mode": {0: "tiled", 1: "bit-parallel, direction planes in HBM", 2: "bit-parallel, checkpoints + replay traceback"}[tm["bit_parallel"]],
                         "note": "frac > 1 is not an accounting error: `achieved` counts the ALGORITHMIC bytes SURVEY 8(d) "
                                 "defines (0.25 B/cell, the 2-bit direction of every cell), but the default checkpoint "
                                 "mode does not write direction planes -- it writes `implementation_bytes_per_launch` "
                                 "(lane-state checkpoints + hand-off marks; equal to the PMC `traffic`) and the traceback "
                                 "replays the blocks on the path. With planes in HBM (CSADP_BITS_CKPT=0) the same metric "
                                 "reads 0.65 and the kernel sits at 88 % of the measured HBM write ceiling (DESIGN.md 3, 5). "
                                 "bytes_per_launch = algorithmic bytes of the `passes_per_launch` passes one launch "
                                 "carries; avg_launch_us = HIP events around ONE such launch alone on its stream (agrees "
                                 "with rocprofv3 --stats on a single stream, profiles/r01_kernel_stats_solo.csv); "
                                 "per_launch_achieved = bytes_per_launch / avg_launch_us; achieved = algorithmic bytes of "
                                 "all timed passes / timed wall time (three launches in flight, so tails and tracebacks "
                                 "hide under the next fill). The binding resource is VALU issue: roofline_valu"},
            "roofline_valu": {"bound": "valu-issue", "ops_per_cell": VALU_OPS_PER_CELL,
                              "issue_cycles_per_64_cells": VALU_ISSUE_CYCLES_PER_CELL_WAVE,
                              "achieved": round(eff_cups / 1e9, 1), "peak": round(VALU_PEAK_CUPS / 1e9, 1),
                              "unit": "GCUPS", "frac": round(eff_cups / VALU_PEAK_CUPS, 4),
                              "note": "peak = 1024 SIMDs x 64 cells / issue cycles x 2.4 GHz with the per-instruction "
                                      "issue rates measured one kind at a time on this chip "
                                      "(profiles/r01_valu_microbench.txt) and the instruction count of the compiled kernel"},
        }
        if args.gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(tasks, results)
            line["cpu_baseline"]["many_cores"] = many_cores
    batch.close()
    group.barrier()
    group.close()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
