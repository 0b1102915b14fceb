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
# VALU issue roofline of the fill kernel.  The benchmarked path is the packed-16 pair kernel
# (nw_fill_tiles_pk): 8 VALU instructions per 2 cells -- v_perm_b32, v_pk_add_i16 x2,
# v_pk_max_i16 x2, v_lshl_add_u32 (half rate: 4 issue cycles per wave64 instruction per SIMD
# when measured one kind at a time) and v_and_b32 x2 (full rate: 2) -- tools/valu_microbench.hip,
# profiles/r01_valu_microbench.txt.  Priced one by one that is 6*4 + 2*2 = 28 issue cycles per
# 128 cells per SIMD -> 11.2 TCUPS at 2.4 GHz: `peak`.  The same microbenchmark runs this exact
# recurrence in registers only (no memory, no tile hand-off): 12.4 TCUPS at 4 waves per SIMD,
# 11.7 at 2 -- reported as `mix_ceiling` (the kernel is capped at 3 waves per SIMD by its 138
# VGPRs).  The 32-bit kernel (N > 2, CSADP_PK16=0): 6 instructions per cell, 8.7 / 6.3 TCUPS.
PK16 = os.environ.get("CSADP_PK16", "1") != "0"
VALU_OPS_PER_CELL = 4 if PK16 else 6
VALU_ISSUE_CYCLES_PER_CELL_WAVE = 14 if PK16 else 18
VALU_PEAK_CUPS = 256 * 4 * 64 / VALU_ISSUE_CYCLES_PER_CELL_WAVE * 2.4e9
VALU_MIX_CEILING_CUPS = 12.36e12 if PK16 else 6.30e12       # cellmix16 / cellmix microbenchmark, 4 waves per SIMD
FILL_KERNEL = "nw_fill_tiles_pk" if PK16 else "nw_fill_tiles"


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
    return {"value": round(cells / secs / 1e9, 4), "unit": "GCUPS", "cores": 1, "kind": kind,
            "sample": "%d of the step's 16 kbp pairs, whole ProgressiveDP (fill+traceback), %.1f s, "
                      "GPU strings %s" % (n, secs, "identical" if mismatches == 0 else "MISMATCH x%d" % mismatches)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--pairs", type=int, default=128, help="pairs per GPU (config 4: 1024 / 8)")
    ap.add_argument("--len", type=int, default=16384, dest="length")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for barriers/reductions; 'gloo' + --share-device rehearses the "
                         "N>1 flow on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="all ranks use HIP device 0 (rehearsal only)")
    args = ap.parse_args()

    import torch
    import csa_amd
    from csa_amd import dist as cdist
    from csa_amd.synth import config4_tasks

    rank, local_rank, world = cdist.env_world()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    dev = 0 if args.share_device else local_rank
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
    tm_pipe = batch.timing()                     # HIP events of the LAST timed pass, on its own stream
    # the same pass once more, alone (no neighbouring pass in flight)
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
        alg_bytes = tm["dir_bytes"] + tm["border_bytes"]       # 0.25 B/cell directions + tile borders, per pass
        # the fill kernel is in flight during the whole timed region (passes overlap on
        # `slots` streams, the traceback of a pass hides under the next pass' fill), so its
        # sustained rate is: work of all timed passes / wall time of the timed region
        rank_cells = tm["cells"]
        eff_bytes_s = alg_bytes * args.steps / elapsed
        eff_cups = rank_cells * args.steps / elapsed
        slots = int(os.environ.get("CSADP_SLOTS", "4"))
        line = {
            "metric": "DP cells/sec (GCUPS) on 16 kbp x 16 kbp pairs, fill + traceback, whole job",
            "value": round(value, 3), "unit": "GCUPS", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16" if PK16 else "int32",
            "data": "synthetic", "verified": bool(ok),
            "per_gpu_gcups": round(value / args.gpus, 3),
            "config": {"workload": "config 4 share: %d synthetic circular %d bp pairs per GPU "
                                   "(global pairs rank*%d..), linear-gap NW fill + traceback, "
                                   "bit-exact vs reference" % (args.pairs, args.length, args.pairs),
                       "pairs_per_gpu": args.pairs, "seq_len": args.length,
                       "cols_per_lane": int(os.environ.get("CSADP_COLS_PER_LANE", "16")),
                       "rows_per_step": int(os.environ.get("CSADP_ROWS_PER_STEP", "2")),
                       "tile_steps": int(os.environ.get("CSADP_TILE_ROWS", "64")),
                       "pipelined_passes": slots,
                       "parallelism": "tasks sharded over %d GPU(s), no collective" % args.gpus},
            "kernel_ms": {"fill_pipelined": round(tm_pipe["fill_ms"], 3),
                          "traceback_pipelined": round(tm_pipe["traceback_ms"], 3),
                          "fill_alone": round(tm["fill_ms"], 3), "traceback_alone": round(tm["traceback_ms"], 3),
                          "fill_launches": tm["fill_launches"], "fill_tiles": tm["fill_tiles"],
                          "fill_alone_gcups": round(tm["cells"] / tm["fill_ms"] / 1e6, 2)},
            "host_boundary_ms": {"create_pack_upload": round(create_s * 1e3, 2), "fetch_download_strings": round(fetch_s * 1e3, 2),
                                 "pcie_inclusive_gcups": round(rank_cells / (create_s + fetch_s + tm["total_ms"] / 1e3) / 1e9, 1),
                                 "note": "not part of value: one batch from host buffers to host strings, unpipelined"},
            "roofline": {"bound": "hbm", "kernel": FILL_KERNEL,
                         "achieved": round(eff_bytes_s / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(eff_bytes_s / 1e9 / HBM_PEAK_GBS, 6),
                         "traffic": pmc_traffic_per_launch(),
                         "bytes_per_launch": round(alg_bytes / launches),
                         "avg_launch_us": round(tm["fill_ms"] * 1e3 / launches, 2),
                         "per_launch_achieved": round(alg_bytes / (tm["fill_ms"] * 1e-3) / 1e9, 1),
                         "stream_us_per_launch_pipelined": round(tm_pipe["fill_ms"] * 1e3 / launches, 2),
                         "concurrent_streams": slots,
                         "note": "algorithmic bytes = 0.25 B/cell directions + tile borders (SURVEY 8d). avg_launch_us = "
                                 "HIP events around the 160 launches of one pass run alone on its stream (agrees with "
                                 "rocprofv3 --stats AverageNs, profiles/r01_kernel_stats*.csv); per_launch_achieved = "
                                 "bytes_per_launch / avg_launch_us. achieved = algorithmic bytes of all timed passes / "
                                 "timed wall time: launches of consecutive passes overlap on `concurrent_streams` streams. "
                                 "dtype: 16-bit lanes relative to exact int32 bases (results bit-exact). The binding "
                                 "roofline is integer VALU issue: roofline_valu"},
            "roofline_valu": {"bound": "valu-issue", "ops_per_cell": VALU_OPS_PER_CELL,
                              "issue_cycles_per_64_cells": VALU_ISSUE_CYCLES_PER_CELL_WAVE,
                              "achieved": round(eff_cups / 1e9, 1), "peak": round(VALU_PEAK_CUPS / 1e9, 1),
                              "unit": "GCUPS", "frac": round(eff_cups / VALU_PEAK_CUPS, 4),
                              "mix_ceiling": round(VALU_MIX_CEILING_CUPS / 1e9, 1),
                              "frac_of_mix_ceiling": round(eff_cups / VALU_MIX_CEILING_CUPS, 4),
                              "note": "peak = 1024 SIMDs x 64 cells / issue cycles x 2.4 GHz with per-op issue rates "
                                      "measured one kind at a time; mix_ceiling = the same recurrence in registers "
                                      "only, as measured on this chip (profiles/r01_valu_microbench.txt)"},
        }
        if args.gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(tasks, results)
    batch.close()
    group.barrier()
    group.close()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
