#!/usr/bin/env python3
"""bench.py -- headline benchmark of the DP hot path on MI355X.

Metric (BASELINE.json): DP cells/sec (GCUPS) on 16 kbp x 16 kbp pairs, 1/2/4/8-GPU batch scaling.

Default workload (every N): each GPU aligns its share of config 4 -- 1024 synthetic circular 16 kbp
pairs over 8 GPUs = 128 pairs per GPU, weak scaling.  A STEP is one pass of the hot path over that
batch starting from the raw circular letters resident in HBM and ending with the two aligned rows
of every pair in HBM: nw_pack_planes (CharAt + letter codes) -> nw_fill_bits (the matrix fill) ->
nw_traceback_windows (the direction walk) -> nw_expand_rows (traceback application + DP score).
One process per GPU; `python bench.py --gpus N` starts its own ranks when it was not launched by
torch.distributed.run.  No collective inside the timed region (independent tasks); afterwards the
16-byte result records of every rank are all-gathered over RCCL and rank 0 checks them.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs P] [--len L]
                    [--mode weak|strong] [--workload config4|config5]

--mode strong: rank 0 owns the whole task list (config 4: 1024 pairs, config 5: 256 mixed-length
pairs), partitions it by LPT (csadp_partition_lpt), broadcasts the assignment; every rank aligns
its part; records are all-gathered, the aligned rows gathered to rank 0, both checked.  --workload mammals: rank 0
reads a real FASTA batch (config 3) and broadcasts it as a packed pool.  Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes
import json
import os
import sys
import time

# the engine's streams (2 fill + 2 side + 1 copy) and torch's share the runtime's hardware queues: ask for 8 before anything
# initialises HIP (csadp_engine.cpp: Engine::open does the same for callers that come to the library first)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Integer VALU peak of the chip: 256 CUs x 4 SIMDs x 32 lanes per clock x 2.4 GHz = 78.6e12 lane
# operations per second (a wave64 instruction issues in 2 cycles on a SIMD-32: MI355X_MICROARCH.md,
# "Wave scheduling"; it is the FP32 vector peak of 157.3 TFLOP/s counted without the FMA's factor 2).
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12
# VALU instructions of the compiled steady-state step of nw_fill_bits per lane (32 cells per word), by words per lane:
# `python tools/count_valu.py csadp_bits.hip nw_fill_bitsILi<W>E...` on the shipped source.  Half-rate on gfx950
# (profiles/r01_valu_microbench.txt): everything that reads another lane (v_sub_co_u32_dpp x3, v_xor_b32_dpp x2) or moves a
# carry (v_addc_co_u32: 3 per word + 3 accumulators).  The calibrated ceiling prices each kind at the issue cost measured
# alone on this chip: v_bitop3 2.6, DPP / carry 4.3, two-operand 2.1 cycles per wave64 and SIMD.
BITS_STEP_MIX = {
    1: {"v_bitop3": (19, 2.6), "dpp_or_carry": (11, 4.3), "two_operand": (1, 2.1)},
    2: {"v_bitop3": (36, 2.6), "dpp_or_carry": (14, 4.3), "two_operand": (2, 2.1)},
    3: {"v_bitop3": (53, 2.6), "dpp_or_carry": (17, 4.3), "two_operand": (3, 2.1)},
    4: {"v_bitop3": (70, 2.6), "dpp_or_carry": (20, 4.3), "two_operand": (4, 2.1)},
}


def bits_valu_per_cell(w):
    return sum(n for n, _ in BITS_STEP_MIX[w].values()) / (32.0 * w)


def bits_cycles_per_64_cells(w):
    return sum(n * c for n, c in BITS_STEP_MIX[w].values()) / (32.0 * w)


ALG_BYTES_PER_CELL = 0.25             # the 2-bit direction of every cell
PREWARM_S = 1.0                       # untimed passes before the W warm-up steps of `value`: the device leaves its idle clocks (main())


PROFILE_ROUNDS = ("r05", "r04")        # the committed rocprofv3 summaries of the newest round that has them


def profile_file(name):
    """profiles/<round>_<name> of the newest round that committed one, or None."""
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", "%s_%s" % (rnd, name))
        if os.path.exists(path):
            return path
    return None


def pmc_summary(kernel):
    """Per-launch PMC figures of `kernel` from the committed rocprofv3 passes (tools/profile_round.sh, tools/summarize_profile.py), or None."""
    try:
        with open(profile_file("pmc_summary.json")) as f:
            return json.load(f)[kernel]
    except Exception:
        return None


def kernel_stats_row(csv_name, kernel_prefix):
    """(calls, average ns) of the first kernel whose name contains `kernel_prefix` in a committed `rocprofv3 --kernel-trace --stats` summary."""
    import csv
    path = profile_file(csv_name)
    if not path:
        return None
    try:
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel_prefix in row["Name"]:
                    return {"file": os.path.relpath(path, ROOT), "kernel": row["Name"].split("(")[0], "calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3}
    except Exception:
        return None
    return None


def roofline_from_profiles(valu_per_cell, launch_cells, wpl):
    """The roofline fraction twice more, each recomputable from ONE committed file under profiles/ (round-4 VERDICT, item 5):
    frac_from_kernel_duration = VALU lane-ops of one launch / rocprofv3's average duration of that launch alone on the device / peak;
    frac_from_counters = SQ_INSTS_VALU x 2 issue cycles / SQ_WAVE_CYCLES of the same launch shape (one wave per SIMD: the wave's own
    share of its SIMD's issue slots).  Both describe a launch ALONE (one wave per SIMD); `frac` beside them is VALU work over the WALL time
    of the pre-warmed timed region with two launches in flight."""
    out = {}
    row = kernel_stats_row("bench_kernel_solo.csv", "nw_fill_bits<%d" % wpl)
    if row:
        tops = valu_per_cell * launch_cells / (row["avg_us"] * 1e-6) / 1e12
        out["kernel_avg_us"] = round(row["avg_us"], 1)
        out["frac_from_kernel_duration"] = round(tops / VALU_PEAK_TOPS, 4)
        out["kernel_duration_source"] = "%s: %s, %d calls" % (row["file"], row["kernel"], row["calls"])
    pmc = pmc_summary("nw_fill_bits")
    if pmc and pmc.get("valu_insts_per_wave") and pmc.get("wave_cycles_per_wave_x4"):
        out["frac_from_counters"] = round(pmc["valu_insts_per_wave"] * 2.0 / pmc["wave_cycles_per_wave_x4"], 4)
        out["counters_source"] = "%s: SQ_INSTS_VALU / SQ_WAVES x 2 / (SQ_WAVE_CYCLES x 4 / SQ_WAVES)" % os.path.relpath(profile_file("pmc_summary.json"), ROOT)
    return out


def host_description():
    """The box the CPU baseline runs on, and how oracle/_ref was built (oracle/Makefile writes build_info.json next to it)."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    build = None
    try:
        with open(os.path.join(ROOT, "oracle", "_ref", "build_info.json")) as f:
            build = json.load(f)
    except Exception:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = None
    return {"nproc": os.cpu_count(), "usable_cores": usable, "cpu_model": model, "reference_build": build}


def reference_faithful_sample(csa_amd, seconds_budget=20.0):
    """SURVEY 8(d)(i) / BASELINE.md section 3: the compiled reference, 1 thread, full int32 matrix + byte directions with one
    malloc per row, on config 2 (Primates 0 x 1 with rotations 1947 / 1949) and the first 8 whole-sequence pairs of Mammals
    (config 3) -- and the GPU's strings for the same tasks beside it."""
    from helpers import GOLDEN, have_ref, load_golden, read_fasta, ref_progressive
    if not have_ref():
        return None
    pipe = load_golden("pipeline.json")
    tasks = []
    names = []
    _, prim = read_fasta(os.path.join(GOLDEN, "data", "Primates.txt"))
    rp = pipe["Primates"]["rotations"]
    tasks.append(([prim[0], prim[1]], [rp[0], rp[1]], None, None))
    names.append("Primates 0x1")
    _, mam = read_fasta(os.path.join(GOLDEN, "data", "Mammals.txt"))
    rm = pipe["Mammals"]["rotations"]
    pairs = [(a, b) for a in range(len(mam)) for b in range(a + 1, len(mam))][:8]
    for a, b in pairs:
        tasks.append(([mam[a], mam[b]], [rm[a], rm[b]], None, None))
        names.append("Mammals %dx%d" % (a, b))
    gpu = csa_amd.align_batch(tasks)
    cells = 0
    secs = 0.0
    done = 0
    same = 0
    first = None
    for t, g in zip(tasks, gpu):
        rc, strs, dt = ref_progressive(t[0], t[1])
        secs += dt
        cells += len(t[0][0]) * len(t[0][1])
        done += 1
        same += 1 if strs == g["aligned"] else 0
        if first is None:
            first = {"pair": names[0], "cells": cells, "seconds": round(dt, 3), "gcups": round(cells / dt / 1e9, 4),
                     "consensus": g["consensus"], "score": g["score"]}
        if secs > seconds_budget:
            break
    return {"value": round(cells / secs / 1e9, 4), "unit": "GCUPS", "cores": 1, "kind": "reference",
            "sample": "config 2 (%s, rotations %d/%d) + the first %d whole-sequence Mammals pairs with the reference's rotations: %d tasks, "
                      "%.1f s inside ProgressiveDP; GPU strings identical for %d of %d" % (names[0], rp[0], rp[1], done - 1, done, secs, same, done),
            "config2": first}


def one_shot_leg(csa_amd, tasks):
    """ONE pass of the step's batch on an idle chip, create -> fetch: what a caller with a single batch of this size gets (the
    timed region keeps several passes in flight and re-uses the resident letters)."""
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        pb = csa_amd.PairBatch(tasks)
        t1 = time.perf_counter()
        pb.run()
        pb.sync()
        t2 = time.perf_counter()
        tm = pb.timing()
        res = pb.fetch()
        t3 = time.perf_counter()
        pb.close()
        ok = all(r["status"] == 0 for r in res)
        cur = {"device_ms": round(tm["total_ms"], 3), "fill_ms": round(tm["fill_ms"], 3), "traceback_expand_ms": round(tm["traceback_ms"], 3),
               "create_ms": round((t1 - t0) * 1e3, 2), "run_and_wait_ms": round((t2 - t1) * 1e3, 2), "fetch_ms": round((t3 - t2) * 1e3, 2),
               "gcups_device": round(tm["cells"] / tm["total_ms"] / 1e6, 1), "gcups_create_to_fetch": round(tm["cells"] / (t3 - t0) / 1e9, 1),
               "words_per_lane": tm["words_per_lane"], "ok": ok, "recoveries": tm["recoveries"]}
        if best is None or cur["device_ms"] < best["device_ms"]:
            best = cur
    best["what"] = ("one pass of the same %d pairs with nothing else on the chip (best of 3): device_ms = HIP events around pack + fill + "
                    "traceback + expand; create_to_fetch = host letters to host strings, Python unpacking included.  A pass flushed alone "
                    "onto an idle device takes the batch's second shape (one word per lane, four strips per workgroup over all compute "
                    "units: csadp_engine.h); round 3 ran it in the steady-state shape: 2.7 ms" % len(tasks))
    return best


def cpu_baseline(tasks, gpu_results, seconds_budget=25.0):
    """Time the CPU path on a bounded sample of the SAME workload on this host, 1 core:
    the compiled reference (oracle/_ref, kind 'reference') when its prebuilt library
    travelled with the repo, else the oracle port.  Also cross-checks the GPU strings."""
    from helpers import have_ref, oracle_progressive, ref_progressive
    keep_heap()   # freed matrix pages stay in the heap between calls (glibc would unmap and re-fault 1.4 GB per pair)
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


def mammals_tasks():
    """The 66 whole-sequence pairs of the reference's Mammals set with the reference's rotations (numpy / file reading only)."""
    from helpers import GOLDEN, load_golden, read_fasta
    _, seqs = read_fasta(os.path.join(GOLDEN, "data", "Mammals.txt"))
    rots = load_golden("pipeline.json")["Mammals"]["rotations"]
    return [([seqs[a], seqs[b]], [rots[a], rots[b]], None, None) for a in range(len(seqs)) for b in range(a + 1, len(seqs))]


def cpu_many_cores(tasks, per_proc=2, o3=False):
    """The same CPU code in P independent processes (the reference keeps global state, so no
    threads), `per_proc` pairs each.  Runs BEFORE this process touches the GPU, so the forked
    children never hold a device context.  o3: the reference built with -O3 -march=znver3
    -fomit-frame-pointer (oracle/Makefile: SURVEY 8(d)(ii)'s second compiler setting)."""
    from helpers import have_ref, have_ref_o3, oracle_progressive, ref_progressive
    if o3 and not have_ref_o3():
        return None
    run = (lambda t, r: ref_progressive(t, r, o3=o3)) if have_ref() else (lambda t, r: oracle_progressive(t, r))
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
            "build": "-O3 -march=znver3 -fomit-frame-pointer" if o3 else "-O2",
            "sample": "%d processes x %d pairs, %.1f s wall" % (procs, per_proc, wall)}


def dropin_leg(sets=("Primates", "Mammals", "Set3")):
    """The boundary north_star names, timed: the reference PROGRAM (mode N) as it is (oracle/_ref/CSA_ref_timed: the unmodified
    sources with a clock around ProgressiveDP, dynamicprogramming.c:906) and relinked with the csadp drop-in in its two modes
    (CSA_csadp: one one-task batch per call, RunAlignment's own pattern alignment.c:179-206; CSA_csadp_deferred: one more link
    flag, all gaps as ONE batch in front of SaveAlignment).  Fresh child processes, started BEFORE this process touches a device.
    The three reference runs go side by side on three host cores (Set3's is 36 fills of up to 17 k x 21 k cells on one core);
    the csadp programs run one at a time with the GPU to themselves, best of two by seconds inside the DP."""
    import hashlib
    import shutil
    import subprocess
    import tempfile
    import threading
    from helpers import GOLDEN, load_golden
    refdir = os.path.join(ROOT, "oracle", "_ref")
    bins = {"reference": "CSA_ref_timed", "synchronous": "CSA_csadp", "deferred": "CSA_csadp_deferred"}
    if not all(os.path.exists(os.path.join(refdir, b)) for b in bins.values()):
        return None
    gold = load_golden("pipeline.json")
    out = {}
    tmp = tempfile.mkdtemp(prefix="csadp_dropin_")

    def start(kind, name):
        d = os.path.join(tmp, "%s_%s_%d" % (kind, name, len(os.listdir(tmp))))
        os.makedirs(d)
        shutil.copy(os.path.join(GOLDEN, "data", name + ".txt"), d)
        env = dict(os.environ)
        env["CSADP_DROPIN_STATS"] = os.path.join(d, "stats.json")
        devnull = open(os.devnull)
        t0 = time.perf_counter()
        p = subprocess.Popen([os.path.join(refdir, bins[kind]), name + ".txt"], cwd=d, stdin=devnull, stdout=subprocess.DEVNULL,
                             stderr=subprocess.DEVNULL, env=env)
        ended = []
        w = threading.Thread(target=lambda: ended.append((p.wait(), time.perf_counter())))     # the process' own end, whenever we look
        w.start()
        return p, d, t0, devnull, w, ended

    def finish(h, name):
        p, d, t0, devnull, w, ended = h
        w.join()
        rc, t1 = ended[0]
        wall = t1 - t0
        devnull.close()
        try:
            with open(os.path.join(d, "stats.json")) as f:
                st = json.loads(f.read().splitlines()[-1])
            with open(os.path.join(d, name + "-Aligned.fasta"), "rb") as f:
                same = hashlib.md5(f.read()).hexdigest() == gold[name]["aligned_md5"]
        except Exception:
            return None
        return {"wall_s": round(wall, 3), "dp_s": round(st["dp_s"], 4), "calls": st["calls"], "batches": st["batches"],
                "init_wait_s": round(st["init_s"], 4), "init_thread_s": round(st.get("early_thread_s", 0.0), 3), "first_call_s": round(st["first_call_s"], 4),
                "aligned_md5_equals_reference": bool(same and rc == 0)}

    try:
        refs = {name: start("reference", name) for name in sets if name in gold}
        for name in refs:
            out[name] = {}
        for name in refs:                                   # the GPU programs meanwhile: one at a time
            if name == "Set3":
                continue
            out[name]["reference"] = finish(refs[name], name)
        for name in refs:
            for kind in ("synchronous", "deferred"):
                best = None
                for _ in range(2):
                    cur = finish(start(kind, name), name)
                    if cur is not None and (best is None or cur["dp_s"] < best["dp_s"]):
                        best = cur
                out[name][kind] = best
        if "Set3" in refs:
            out["Set3"]["reference"] = finish(refs["Set3"], "Set3")
        for name in refs:
            r, sy, de = out[name].get("reference"), out[name].get("synchronous"), out[name].get("deferred")
            if r and sy and de:
                out[name]["dp_speedup_vs_reference"] = {"synchronous": round(r["dp_s"] / sy["dp_s"], 1), "deferred": round(r["dp_s"] / de["dp_s"], 1)}
                out[name]["wall_speedup_vs_reference"] = {"synchronous": round(r["wall_s"] / sy["wall_s"], 2), "deferred": round(r["wall_s"] / de["wall_s"], 2)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out["what"] = ("mode N of the reference program on its example sets: `reference` = unmodified sources (1 host core), `synchronous` / `deferred` = "
                   "the same program with dynamicprogramming.c replaced by csadp_dropin.c (+ -Wl,--wrap=SaveAlignment for deferred).  dp_s = seconds "
                   "inside ProgressiveDP calls (+ the one batch of the deferred mode); init_thread_s = csadp_init + csadp_warmup (HIP start-up, code objects, arenas) on the "
                   "adapter's helper thread, started when the program starts, under the program's own suffix-tree stage; init_wait_s = what "
                   "the first gap still waits for it; neither is in dp_s; wall_s = the whole process, the reference's own single-threaded suffix tree and anchor stages included")
    return out


def streaming_leg(csa_amd, tasks, batches, depth=3):
    """What a caller of the C-ABI gets from host buffers to host strings, PCIe included: `batches`
    pair batches, `depth` in flight (create + run + flush of batch n+depth-1 before fetch of batch n).
    Only C-ABI calls inside the timed loop (the ctypes task arrays are built before it)."""
    L = csa_amd.lib()
    ta = csa_amd.TaskArray(tasks)
    res = [(csa_amd.Result * ta.n)() for _ in range(depth)]
    cells = sum(len(t[0][0]) * len(t[0][1]) for t in tasks)

    def start():
        h = ctypes.c_void_p()
        rc = L.csadp_pairs_create(ta.arr, ta.n, ctypes.byref(h))
        rc = rc or L.csadp_pairs_run(h) or L.csadp_pairs_flush(h)
        if rc:
            raise csa_amd.CsadpError(rc, "streaming leg")
        return h

    def finish(h, slot):
        rc = L.csadp_pairs_fetch(h, res[slot])
        if rc:
            raise csa_amd.CsadpError(rc, "streaming leg fetch")
        failed = L.csadp_free_results(res[slot], ta.n, 2)       # one call: 512 ctypes calls per batch cost 2.7 ms of the loop
        L.csadp_pairs_destroy(h)
        return failed == 0

    ok = True
    wall = 0.0
    for phase_batches in (depth + 1, batches):       # warm-up (fills the engine's buffer pools), then timed
        inflight = []
        t0 = time.perf_counter()
        for n in range(phase_batches):
            inflight.append(start())
            if len(inflight) == depth:
                ok = finish(inflight.pop(0), n % depth) and ok
        while inflight:
            ok = finish(inflight.pop(0), 0) and ok
        wall = time.perf_counter() - t0
    return {"gcups": round(cells * batches / wall / 1e9, 1), "ms_per_batch": round(wall * 1e3 / batches, 3),
            "batches": batches, "in_flight": depth, "ok": ok,
            "what": "host letters -> H2D -> pack, fill, traceback, expand -> D2H -> malloc'd result strings, "
                    "several batches in flight through csadp_pairs_create/run/flush/fetch"}


def timed_pair_batch(csa_amd, tasks, steps, warmup):
    """`steps` passes of one device-resident pair batch, timed like the headline (letters in HBM -> aligned rows in HBM; every
    pass enqueued, then one wait), after `warmup` untimed ones.  Returns (seconds, timing record of the last launch, results)."""
    pb = csa_amd.PairBatch(tasks)
    pb.sync()
    for _ in range(warmup):
        pb.run()
    pb.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        pb.run()
    pb.sync()
    dt = time.perf_counter() - t0
    tm = pb.timing()
    res = pb.fetch()
    pb.close()
    return dt, tm, res


def real_sets_leg(csa_amd, steps=48, warmup=8):
    """BASELINE config 3 and its second point: ALL whole-sequence pairs of the reference's example sets (Manual/Mammals.txt: 66,
    Manual/Primates.txt: 120) with the reference's rotations, as one device-resident batch each; every record (SP score,
    consensus, FNV-1a of the two rows) against the compiled reference's (tests/golden/real_pairs.json)."""
    from helpers import GOLDEN, load_golden, read_fasta
    out = {}
    gold_all = load_golden("real_pairs.json")
    pipe = load_golden("pipeline.json")
    for setname in ("Mammals", "Primates"):
        _, seqs = read_fasta(os.path.join(GOLDEN, "data", setname + ".txt"))
        rots = pipe[setname]["rotations"]
        pair_of = [(a, b) for a in range(len(seqs)) for b in range(a + 1, len(seqs))]
        tasks = [([seqs[a], seqs[b]], [rots[a], rots[b]], None, None) for a, b in pair_of]
        dt, tm, res = timed_pair_batch(csa_amd, tasks, steps, warmup)
        gold = {(c["a"], c["b"]): c for c in gold_all if c["set"] == setname and list(c["rots"]) == [rots[c["a"]], rots[c["b"]]]}
        same = 0
        for (a, b), r in zip(pair_of, res):
            g = gold.get((a, b))
            if g and r["status"] == 0 and (r["score"], r["consensus"], csa_amd.fnv1a(r["aligned"])) == (g["sp"], g["consensus"], int(g["fnv1a"], 16)):
                same += 1
        out[setname] = {"pairs": len(tasks), "cells_per_step": tm["cells"], "gcups": round(tm["cells"] * steps / dt / 1e9, 1),
                        "ms_per_step": round(dt * 1e3 / steps, 3), "steps": steps, "warmup": warmup,
                        "equal_to_reference_digests": same, "words_per_lane": tm["words_per_lane"],
                        "passes_per_launch": tm["merge_group"], "launches_in_flight": tm["streams"], "recoveries": tm["recoveries"]}
    out["what"] = ("config 3: the 66 Mammals and the 120 Primates whole-sequence pairs (16.3-17.7 k letters, gap-rich paths), letters in HBM -> "
                   "aligned rows in HBM, timed like `value`")
    return out


def config5_leg(csa_amd, steps=3, warmup=1):
    """BASELINE config 5: 256 pairs of 1-200 kbp (1.2e12 cells per pass) as ONE device-resident batch; every result re-spells its
    inputs and scores what its rows score."""
    from csa_amd.synth import config5_lengths, synth_pair
    from helpers import degap, rotated, sp_score
    la, _ = config5_lengths(256)
    tasks = []
    for i, length in enumerate(la):
        a, b, ra, rb = synth_pair(20000 + i, length=int(length))
        tasks.append(([a, b], [ra, rb], None, None))
    dt, tm, res = timed_pair_batch(csa_amd, tasks, steps, warmup)
    ok = True
    for t, r in zip(tasks, res):
        ok = ok and r["status"] == 0 and degap(r["aligned"][0]) == rotated(t[0][0], t[1][0]) and degap(r["aligned"][1]) == rotated(t[0][1], t[1][1])
        ok = ok and sp_score(r["aligned"]) == r["score"]
    # every pair the reference's 5 B/cell matrices can hold here (both sides <= 40 000 letters: 171 of the 256) against the compiled reference
    from helpers import load_golden
    gold = load_golden("config5_pairs.json")
    same = sum(1 for g in gold if (res[g["index"]]["score"], res[g["index"]]["consensus"], csa_amd.fnv1a(res[g["index"]]["aligned"])) ==
               (g["sp"], g["consensus"], int(g["fnv1a"], 16)))
    return {"pairs": len(tasks), "cells_per_step": tm["cells"], "gcups": round(tm["cells"] * steps / dt / 1e9, 1),
            "ms_per_step": round(dt * 1e3 / steps, 2), "steps": steps, "warmup": warmup, "words_per_lane": tm["words_per_lane"],
            "properties_hold_for_all": bool(ok), "reference_digests": len(gold), "equal_to_reference_digests": same, "recoveries": tm["recoveries"],
            "what": "256 synthetic pairs of 1-200 kbp in one batch (jobs of up to 33 strips run as chains of workgroups)"}


def unrelated_leg(csa_amd, steps=20, warmup=5):
    """SURVEY 8(d), config 4's second variant: 64 UNRELATED 16 kbp pairs -- the gap-richest paths a 16 kbp pair has."""
    from csa_amd.synth import synth_pair
    from helpers import degap, rotated, sp_score
    tasks = []
    for p in range(64):
        a, b, ra, rb = synth_pair(70000 + p, unrelated=True)
        tasks.append(([a, b], [ra, rb], None, None))
    dt, tm, res = timed_pair_batch(csa_amd, tasks, steps, warmup)
    ok = True
    for t, r in zip(tasks, res):
        ok = ok and r["status"] == 0 and degap(r["aligned"][0]) == rotated(t[0][0], t[1][0]) and degap(r["aligned"][1]) == rotated(t[0][1], t[1][1])
        ok = ok and sp_score(r["aligned"]) == r["score"]
    from helpers import load_golden
    gold = load_golden("unrelated_pairs.json")
    same = sum(1 for g in gold if (res[g["pair"]]["score"], res[g["pair"]]["consensus"], csa_amd.fnv1a(res[g["pair"]]["aligned"])) ==
               (g["sp"], g["consensus"], int(g["fnv1a"], 16)))
    return {"pairs": len(tasks), "gcups": round(tm["cells"] * steps / dt / 1e9, 1), "ms_per_step": round(dt * 1e3 / steps, 3),
            "steps": steps, "warmup": warmup, "words_per_lane": tm["words_per_lane"], "properties_hold_for_all": bool(ok),
            "equal_to_reference_digests": same, "recoveries": tm["recoveries"],
            "what": "64 unrelated random 16384-letter pairs, every record against the compiled reference's (tests/golden/unrelated_pairs.json)"}


def small_pairs_leg(csa_amd, steps=48, warmup=8):
    """Batches of pairs narrower than a four-strip workgroup (what the anchors of the reference's pipeline leave between them when a set is
    aligned pair by pair): 256 pairs of 8 192 letters (two strips at two words per lane), 128 of 12 000 (two at three) and 256 of mixed
    lengths (500-16 000: one to four strips), with the jobs sharing four-wave workgroups of nw_fill_bits by first fit (the default) and one
    workgroup per job (CSADP_BITS_PACK=0); properties checked on every result."""
    from csa_amd.synth import synth_pair
    from helpers import degap, rotated, sp_score
    out = {}
    import random
    mixed = random.Random(5)
    for name, npairs, length in (("256_pairs_of_8192", 256, 8192), ("128_pairs_of_12000", 128, 12000), ("256_pairs_of_500_to_16000", 256, 0)):
        tasks = []
        for p in range(npairs):
            a, b, ra, rb = synth_pair(90000 + p, length=length or mixed.randrange(500, 16000))
            tasks.append(([a, b], [ra, rb], None, None))
        entry = {}
        for tag, pack in (("shared_workgroups", None), ("one_workgroup_per_job", "0")):
            if pack is not None:
                os.environ["CSADP_BITS_PACK"] = pack
            csa_amd.reload_config()
            dt, tm, res = timed_pair_batch(csa_amd, tasks, steps, warmup)
            os.environ.pop("CSADP_BITS_PACK", None)
            csa_amd.reload_config()
            ok = all(r["status"] == 0 and degap(r["aligned"][0]) == rotated(t[0][0], t[1][0]) and degap(r["aligned"][1]) == rotated(t[0][1], t[1][1]) and
                     sp_score(r["aligned"]) == r["score"] for t, r in zip(tasks, res))
            entry[tag] = {"gcups": round(tm["cells"] * steps / dt / 1e9, 1), "ms_per_step": round(dt * 1e3 / steps, 3), "words_per_lane": tm["words_per_lane"],
                          "passes_per_launch": tm["merge_group"], "launches_in_flight": tm["streams"], "properties_hold_for_all": bool(ok),
                          "recoveries": tm["recoveries"]}
        out[name] = entry
    out["what"] = "pair jobs narrower than four strips: jobs sharing four-wave workgroups (first fit) against one workgroup per job, 48 + 8 steps each"
    return out


def config4_all_leg(csa_amd, first_tasks, steps=4, warmup=1):
    """BASELINE config 4 whole: all 1024 synthetic pairs on this one GPU as ONE device-resident batch (what the 8 ranks of the
    scaling run share out 128 apiece), every record against the compiled reference's (tests/golden/config4_all.json)."""
    from csa_amd.synth import config4_tasks
    from helpers import load_golden
    tasks = list(first_tasks) + config4_tasks(len(first_tasks), 1024 - len(first_tasks))
    dt, tm, res = timed_pair_batch(csa_amd, tasks, steps, warmup)
    gold = load_golden("config4_all.json")
    same = sum(1 for p, r in enumerate(res) if r["status"] == 0 and
               (r["score"], r["consensus"], csa_amd.fnv1a(r["aligned"])) == (gold["sp"][p], gold["consensus"][p], int(gold["fnv1a"][p], 16)))
    return {"pairs": len(tasks), "cells_per_step": tm["cells"], "gcups": round(tm["cells"] * steps / dt / 1e9, 1),
            "ms_per_step": round(dt * 1e3 / steps, 3), "steps": steps, "warmup": warmup, "words_per_lane": tm["words_per_lane"],
            "equal_to_reference_digests": same, "recoveries": tm["recoveries"],
            "what": "all 1024 pairs of config 4 in one batch on one GPU; every (score, consensus, FNV-1a of the rows) equal to the compiled reference's"}


def keep_heap():
    """glibc maps and unmaps every vector of more than 128 KB: the host stages of mode N (suffix automata of a few MB per sequence) and the
    CPU baseline (1.4 GB of matrices per pair) re-fault their pages at every call.  A program that calls them in a loop keeps the heap
    (INTEGRATION.md, section 2); the library itself leaves the process' malloc settings alone."""
    try:
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-1, 1 << 30)   # M_TRIM_THRESHOLD
        libc.mallopt(-3, 1 << 25)   # M_MMAP_THRESHOLD (32 MiB is the glibc maximum)
        return True
    except Exception:
        return False


def profile_path_leg(csa_amd):
    """The reference's OWN use of ProgressiveDP (mode N): sequence-vs-profile fills (i up to 18) of the
    example sets through csadp_msa -- one device batch per set, lock-step rounds.  Best of 3 warm calls (by DP time; `total_ms_best` = the
    shortest whole call of the three)."""
    from helpers import GOLDEN, read_fasta
    out = {}
    kept = keep_heap()
    for name in ("Primates", "Mammals", "Set3"):
        path = os.path.join(GOLDEN, "data", name + ".txt")
        if not os.path.exists(path):
            continue
        _, seqs = read_fasta(path)
        best = None
        walls = []
        for _ in range(3):
            t0 = time.perf_counter()
            rc, rot, rows, st = csa_amd.msa(seqs)
            wall = time.perf_counter() - t0
            if rc != 0:
                best = None
                break
            walls.append(wall)
            if best is None or st["dp_ms"] < best["dp_ms"]:
                best = {"dp_ms": round(st["dp_ms"], 2), "total_ms": round(wall * 1e3, 1), "fills": st["fills"],
                        "gaps": st["dp_gaps"], "cells": st["cells"], "gcups": round(st["cells"] / st["dp_ms"] / 1e6, 1),
                        "recoveries": st.get("recoveries", 0)}
        if best is not None:
            best["total_ms_best"] = round(min(walls) * 1e3, 1)
        out[name] = best
    out["host_heap"] = "kept (mallopt M_TRIM_THRESHOLD 1 GiB, M_MMAP_THRESHOLD 32 MiB in this process)" if kept else "glibc defaults"
    return out


def profile_batch_leg(csa_amd):
    """Batches of N-SEQUENCE tasks through csadp_align_batch: profile fills (dynamicprogramming.c:990-1029 with i >= 2) at full occupancy --
    nw_fill_cells in launches of a thousand and more workgroups -- which the reference's own sets never produce (their rounds are a handful of
    matrices).  End to end, and where the time goes (csadp_last_batch_phases): the device part, and the host's part of ProgressiveDP between two
    fills (trace application :1050-1155, DeleteGappedColumns :643-899, the next fill's tables).  Families are co-linear (what lies between two
    anchors of the reference's pipeline); the `misrotated` entry gives every sequence a random rotation instead -- alignments that are mostly end
    gaps, the worst case of DeleteGappedColumns, where the rounds become host-bound.  Best of two calls after a warm one."""
    from helpers import random_family, rng
    out = {}
    for name, nfam, nseq, length, rotate in (("256_families_of_8x4000", 256, 8, 4000, False), ("16_families_of_16x16000", 16, 16, 16000, False),
                                             ("64_families_of_4x30000", 64, 4, 30000, False),
                                             ("64_families_of_8x4000_misrotated", 64, 8, 4000, True)):
        r = rng(nfam * 1000 + nseq)
        tasks = []
        for _ in range(nfam):
            fam = random_family(r, nseq, length, mut=0.08, indel=0.02)
            tasks.append((fam, [r.randrange(len(x)) for x in fam] if rotate else None, None, None))
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            got = csa_amd.align_batch(tasks)
            dt = time.perf_counter() - t0
            ph = csa_amd.last_batch_phases()
            cells = sum(g["cells"] for g in got)
            ok = all(g["status"] == 0 and len(set(len(x) for x in g["aligned"])) == 1 for g in got)
            cur = {"ms": round(dt * 1e3, 1), "gcups": round(cells / dt / 1e9, 1), "cells": cells, "fills": sum(g["fills"] for g in got),
                   "rounds": ph["rounds"], "round_groups": ph["round_groups"], "ok": bool(ok),
                   "phases_ms_summed_over_rounds_and_groups": {k: round(ph[k], 1) for k in (
                       "device_ms", "tables_ms", "apply_ms", "refine_speculate_ms", "refine_commit_ms", "seed_ms", "results_ms")},
                   "recoveries": csa_amd.recoveries()}
            if rep > 0 and (best is None or cur["ms"] < best["ms"]):
                best = cur
        out[name] = best
    kr = kernel_stats_row("pbatch_kernel_stats.csv", "nw_fill_cells")
    if kr:
        out["nw_fill_cells_under_rocprofv3"] = dict(kr, what="launches of ~580 workgroups (64 matrices of 4 000 x 4 050, one round group's round of the 256-family batch), "
                                                            "up to four such launches on the chip at once: the duration is a launch's own, not a solo rate")
    out["what"] = ("families of random related sequences through csadp_align_batch: lock-step rounds over two round groups, four from 128 tasks on; 8 x 4 kbp: 9 workgroups of "
                   "nw_fill_cells per matrix; 16 x 16 kbp: 32-40 per matrix.  Co-linear families run device-bound (the two groups' "
                   "device parts overlap); misrotated ones are bound by DeleteGappedColumns on the host (DESIGN.md section 10).  nw_fill_cells is issue-bound at 86 cycles per "
                   "step of 128 cells per SIMD whatever its occupancy: 3.66 TCUPS for the chip (DESIGN.md section 3 K1c); the first fill of every family runs on nw_fill_bits")
    return out


def single_matrix_leg(csa_amd):
    """Latency of ONE matrix (BASELINE config 2: a 16 kbp pair; config 5's upper end: 100 and 200 kbp pairs): device fill and
    traceback of a one-job batch, best of 3.  A large pair alone takes the cell-per-lane kernels (every chunk of the matrix on a
    compute unit of its own, band-parallel walk; FillBatch::lone_pairs_take_cells) while its chunks number at most 256: `path`
    says which kernels ran (200 kbp: 391 chunks, bit-parallel)."""
    from csa_amd.synth import synth_pair
    out = {}
    for length in (16384, 100000, 200000):
        a, b, ra, rb = synth_pair(777, length=length)
        pb = csa_amd.PairBatch([([a, b], [ra, rb], None, None)])
        best = None
        for _ in range(3):
            pb.run()
            pb.sync()
            t = pb.timing()
            if best is None or t["total_ms"] < best["total_ms"]:
                best = t
        pb.fetch()
        pb.close()
        out[str(length)] = {"fill_ms": round(best["fill_ms"], 3), "traceback_expand_ms": round(best["traceback_ms"], 3),
                            "gcups": round(len(a) * len(b) / best["total_ms"] / 1e6, 1),
                            "path": "cell-per-lane" if best["words_per_lane"] == 0 else "bit-parallel", "recoveries": best["recoveries"]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=128, help="pairs per GPU (config 4: 1024 / 8)")
    ap.add_argument("--len", type=int, default=16384, dest="length")
    ap.add_argument("--mode", default="weak", choices=["weak", "strong"])
    ap.add_argument("--workload", default="config4", choices=["config4", "config5", "mammals", "primates"],
                    help="--mode strong: which list rank 0 owns; mammals = config 3, all 66 whole-sequence pairs of "
                         "tests/golden/data/Mammals.txt with the reference's rotations: rank 0 reads the FASTA and broadcasts the pool")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the streaming and profile-path legs")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; 'gloo' + --share-device rehearses the N>1 flow on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="all ranks use HIP device 0 (rehearsal only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # invoked plainly: start one child per GPU before anything here has touched a device
        from csa_amd.dist import spawn_ranks
        raise SystemExit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    # stdout carries ONE line, the JSON of rank 0: whatever libraries write there on the way (gloo announces its connections on
    # stdout) is sent to stderr, the line itself goes to the descriptor kept aside
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    many_cores = many_cores_o3 = many_cores_c3 = dropin = None
    if args.gpus == 1 and not args.no_cpu_baseline and not args.no_extra_legs:
        dropin = dropin_leg()
    if args.gpus == 1 and not args.no_cpu_baseline:
        from csa_amd.synth import config4_tasks as _tasks      # numpy only: no device is initialised here
        many_cores = cpu_many_cores(_tasks(0, min(args.pairs, 64), args.length))
        many_cores_o3 = cpu_many_cores(_tasks(0, min(args.pairs, 64), args.length), o3=True)
        many_cores_c3 = cpu_many_cores(mammals_tasks()[:64])       # configs 3 AND 4
        if many_cores_c3:
            many_cores_c3["sample"] = "config 3 (Mammals whole-sequence pairs): " + many_cores_c3["sample"]

    import csa_amd
    from csa_amd import dist as cdist
    from csa_amd.synth import config4_tasks, config5_lengths, synth_pair

    rank, local_rank, world = cdist.env_world()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev = 0 if args.share_device else local_rank
    if args.share_device:
        os.environ["CSADP_SHARE_DEVICE"] = "1"
    torch = None
    if args.backend == "nccl":
        import torch
        torch.cuda.set_device(dev)
    csa_amd.init(device=dev)
    group = cdist.Group(backend=args.backend, device="cuda:%d" % dev)

    imbalance = 1.0
    part = None
    if args.mode == "weak":
        # rank r owns global pairs [r*P, (r+1)*P) of the synthetic batch
        ids = list(range(rank * args.pairs, (rank + 1) * args.pairs))
        tasks = config4_tasks(rank * args.pairs, args.pairs, args.length)
        workload = "config 4 share: %d synthetic circular %d bp pairs per GPU (global pairs rank*%d ..)" % (
            args.pairs, args.length, args.pairs)
    else:
        # rank 0 owns the list; cost = DP cells from the nominal lengths; LPT + broadcast
        pair_of = None
        if args.workload in ("mammals", "primates"):
            # config 3 (and its second point, the 120 Primates pairs): rank 0 owns the FASTA batch; every other rank receives
            # the letters and rotations as one packed pool
            setname = args.workload.capitalize()
            seqs = rots = None
            if rank == 0:
                from helpers import GOLDEN, load_golden, read_fasta
                _, seqs = read_fasta(os.path.join(GOLDEN, "data", setname + ".txt"))
                rots = load_golden("pipeline.json")[setname]["rotations"]
            pool, prot = group.broadcast_pool(seqs, rots)
            pair_of = [(a, b) for a in range(len(pool)) for b in range(a + 1, len(pool))]
            lens = [(len(pool[a]), len(pool[b])) for a, b in pair_of]

            def make(p):
                a, b = pair_of[p]
                return pool[a], pool[b], prot[a], prot[b]
        elif args.workload == "config4":
            lens = [(args.length, args.length)] * 1024

            def make(p):
                return synth_pair(p, args.length)
        else:
            la, lb = config5_lengths(256)
            lens = list(zip(la, lb))

            def make(p):
                return synth_pair(20000 + p, length=int(la[p]))
        part = cdist.lpt_assignment(group, [a * b for a, b in lens])
        ids = [t for t, p in enumerate(part) if p == rank]
        load = [0] * world
        for t, p in enumerate(part):
            load[p] += lens[t][0] * lens[t][1]
        imbalance = max(load) * world / sum(load)
        tasks = []
        for p in ids:
            a, b, ra, rb = make(p)
            tasks.append(([a, b], [ra, rb], None, None))
        workload = "%s, all %d pairs partitioned by LPT over the ranks" % (args.workload, len(lens))

    t_c0 = time.perf_counter()
    batch = csa_amd.PairBatch(tasks)            # argument checks, letters -> pinned memory, H2D started
    batch.sync()
    create_s = time.perf_counter() - t_c0

    def sync():
        batch.sync()
        if torch is not None:
            torch.cuda.synchronize()

    # The chip holds its clocks down after an idle period and takes a few hundred milliseconds of load to raise them
    # (MI355X_MICROARCH.md, 'DVFS give-back': steady clocks want ~2 s of back-to-back launches; tools/r04/prewarm_probe.py,
    # profiles/r04_prewarm_probe.txt: W = 5 warm-up passes are 4 ms -- 42.9-43.5 TCUPS from idle clocks, 46-47 after 0.5-2 s of
    # load, 49.8 sustained over 1200 steps).  So: first the driver's W + K steps as they come, from idle clocks (reported as
    # `from_idle_clocks`), then PREWARM_S seconds of untimed passes of the same batch, then the W warm-up and EXACTLY K timed steps
    # of `value`.
    elapsed_idle = cdist.timed_steps(group, batch.run, sync, args.steps, args.warmup)
    t_pw = time.perf_counter()
    prewarm_passes = 0
    while time.perf_counter() - t_pw < PREWARM_S:
        for _ in range(16):
            batch.run()
        sync()
        prewarm_passes += 16
    elapsed = cdist.timed_steps(group, batch.run, sync, args.steps, args.warmup)
    tm_pipe = batch.timing()                     # HIP events of the launch that held the LAST timed pass
    # one FULL launch alone (nothing else in flight): the bit-parallel path merges up to
    # `passes_per_launch` consecutive passes into one launch, whatever --steps is
    for _ in range(max(tm_pipe["merge_group"], 1)):
        batch.run()
    sync()
    tm = batch.timing()
    cells_step = group.sum(tm["cells"])
    t_f0 = time.perf_counter()
    results = batch.fetch()                      # ONE D2H of the aligned rows + malloc'd result strings
    fetch_s = time.perf_counter() - t_f0
    ok = all(r["status"] == 0 for r in results)

    # the multi-GPU data path: every rank's 16-byte records (task id, score, consensus, FNV-1a) all-gathered, and the aligned
    # rows themselves -- what the reference's consumer prints (alignment.c:134-156) -- gathered to rank 0
    recs = group.all_gather_records([(t, r["score"], r["consensus"], csa_amd.fnv1a(r["aligned"])) for t, r in zip(ids, results)])
    by_id = {r[0]: (r[1], r[2], cdist.u32(r[3])) for r in recs}
    t_g0 = time.perf_counter()
    rows_by_id = group.gather_rows(ids, [r["aligned"] for r in results])
    gather_s = time.perf_counter() - t_g0

    value = cells_step * args.steps / elapsed / 1e9
    line = None
    if rank == 0:
        from helpers import degap, load_golden, rotated, sp_score
        # properties on this rank's first pairs: strings re-spell the inputs, SP == DP score
        for t, r in list(zip(tasks, results))[:4]:
            ok = ok and degap(r["aligned"][0]) == rotated(t[0][0], t[1][0]) and degap(r["aligned"][1]) == rotated(t[0][1], t[1][1])
            ok = ok and sp_score(r["aligned"]) == r["score"]
        # gathered records: complete, and the ones the compiled reference has digests for agree with it
        expect = args.pairs * world if args.mode == "weak" else len(part)
        ok = ok and len(by_id) == expect
        # every task's rows have arrived on rank 0 and are the rows their record was computed from
        rows_ok = len(rows_by_id) == expect and all(csa_amd.fnv1a(rows_by_id[t]) == by_id[t][2] for t in by_id)
        ok = ok and rows_ok
        checked = 0
        if args.mode == "strong" and args.workload in ("mammals", "primates"):
            gold = {(c["a"], c["b"]): c for c in load_golden("real_pairs.json")
                    if c["set"] == args.workload.capitalize() and list(c["rots"]) == [prot[c["a"]], prot[c["b"]]]}
            for t, (a, b) in enumerate(pair_of):
                g = gold[(a, b)]
                ok = ok and by_id[t] == (g["sp"], g["consensus"], int(g["fnv1a"], 16))
                checked += 1
        if args.workload == "config4" and args.length == 16384:
            g4 = load_golden("config4_all.json")             # all 1024 pairs through the compiled reference
            for p in by_id:
                if p < g4["pairs"]:
                    ok = ok and by_id[p] == (g4["sp"][p], g4["consensus"][p], int(g4["fnv1a"][p], 16))
                    checked += 1
        lp = max(tm["launch_passes"], 1)                        # passes carried by the launch timed alone
        impl_bytes = tm["dir_bytes"] + tm["border_bytes"]      # what the fill writes per pass (checkpoints + marks, or planes)
        rank_cells = tm["cells"]
        launch_cells = lp * rank_cells
        fill_s = tm["fill_ms"] * 1e-3
        wpl = tm["words_per_lane"] if tm["words_per_lane"] in BITS_STEP_MIX else 1
        valu_per_cell = bits_valu_per_cell(wpl)
        alone_tops = valu_per_cell * launch_cells / fill_s / 1e12               # lane operations per second
        sustained_tops = valu_per_cell * rank_cells * args.steps / elapsed / 1e12
        calibrated_peak_gcups = 256 * 4 * 64 / bits_cycles_per_64_cells(wpl) * 2.4
        pmc = pmc_summary("nw_fill_bits")
        traffic = None
        if pmc and pmc.get("launch_shape") == {"jobs": lp * len(tasks), "len": args.length}:
            traffic = int(pmc["hbm_write_bytes_per_launch"] + pmc["hbm_read_bytes_per_launch_x2_corrected"])
        line = {
            "metric": "DP cells/sec (GCUPS) on %s, letters in HBM -> aligned rows in HBM, whole job" % (
                "1-200 kbp mixed-length pairs (config 5)" if args.workload == "config5" and args.mode == "strong" else
                "the 66 whole-sequence Mammals pairs (config 3)" if args.workload == "mammals" and args.mode == "strong" else
                "the 120 whole-sequence Primates pairs (config 3, second point)" if args.workload == "primates" and args.mode == "strong" else
                "16 kbp x 16 kbp pairs"),
            "value": round(value, 3), "unit": "GCUPS", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 3),
            "higher_is_better": True, "scaling": args.mode, "vs_baseline": None,
            "dtype": "u32 bit planes",
            "data": "synthetic", "verified": bool(ok),
            "per_gpu_gcups": round(value / args.gpus, 3),
            "clocks": {"prewarm_s": PREWARM_S, "prewarm_passes": prewarm_passes,
                       "from_idle_clocks": {"value": round(cells_step * args.steps / elapsed_idle / 1e9, 3), "ms_per_step": round(elapsed_idle * 1e3 / args.steps, 3),
                                            "what": "the same W warm-up + K timed steps run FIRST, before any other load on the device"},
                       "what": "`value` is timed after %.1f s of untimed passes of the same batch (then the W warm-up steps, then exactly K steps): the chip "
                               "takes a few hundred ms of load to leave its idle clocks, W = 5 passes are 4 ms (MI355X_MICROARCH.md 'DVFS give-back'; "
                               "profiles/r04_prewarm_probe.txt: 42.9-43.5 TCUPS from idle, 46-47 after 0.5-2 s of load, 49.8 sustained over 1200 steps)" % PREWARM_S},
            "config": {"workload": workload + "; linear-gap NW: pack + fill + traceback + row expansion on the device, "
                                              "bit-exact vs the reference",
                       "pairs_this_rank": len(tasks), "seq_len": "1000..200000" if args.workload == "config5" and args.mode == "strong" else args.length,
                       "kernel": "nw_fill_bits", "words_per_lane": tm["words_per_lane"], "launches_in_flight": tm_pipe["streams"],
                       "passes_per_launch": max(tm_pipe["launch_passes"], 1), "device_io": tm["device_io"],
                       "parallelism": "independent tasks over %d GPU(s); no collective in the timed region; result records "
                                      "all-gathered over %s afterwards" % (args.gpus, "RCCL" if args.backend == "nccl" else "gloo"),
                       "lpt_imbalance": round(imbalance, 4),
                       # the clock story where the driver's parser keeps it: `value` follows PREWARM_S seconds of untimed passes;
                       # the same W + K steps from idle clocks (what rounds 1-3 reported as `value`) are value_from_idle_gcups
                       "prewarm_s": PREWARM_S, "value_from_idle_gcups": round(cells_step * args.steps / elapsed_idle / 1e9, 3)},
            "records": {"gathered": len(by_id), "checked_against_reference_digests": checked,
                        "rows_on_rank0": len(rows_by_id), "rows_match_their_records": bool(rows_ok),
                        "rows_bytes": sum(len(x) for v in rows_by_id.values() for x in v), "rows_gather_ms": round(gather_s * 1e3, 2)},
            "kernel_ms": {"fill_launch_alone": round(tm["fill_ms"], 3),
                          "traceback_and_expand_alone": round(tm["traceback_ms"], 3),
                          "passes_in_that_launch": lp,
                          "fill_launch_pipelined": round(tm_pipe["fill_ms"], 3),
                          "fill_alone_gcups": round(launch_cells / tm["fill_ms"] / 1e6, 1),
                          "recoveries": tm["recoveries"]},
            "host_boundary": {"create_ms": round(create_s * 1e3, 2), "fetch_ms": round(fetch_s * 1e3, 2),
                              "note": "not part of value; create = argument checks + letters to pinned memory + H2D, "
                                      "fetch = one D2H + result strings (Python unpacking included here; the C-ABI "
                                      "figure is streaming.ms_per_batch)"},
            "roofline": {"bound": "valu-issue", "kernel": "nw_fill_bits",
                         "achieved": round(sustained_tops, 2), "peak": round(VALU_PEAK_TOPS, 2), "unit": "TOP/s",
                         "frac": round(sustained_tops / VALU_PEAK_TOPS, 4),
                         "traffic": traffic,
                         "what": "integer VALU lane-operations per second of the fill kernel over the timed region: %d wave64 "
                                 "VALU instructions per lane-step of %d cells (ISA count of the compiled step, %d words per lane: "
                                 "tools/count_valu.py) x cells of all timed passes / wall time -- %d launches of `passes_per_launch` "
                                 "passes are in flight on as many streams, each launch's traceback on a side stream, so the kernel runs "
                                 "during the whole region; peak = 256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz (nominal 2 issue cycles "
                                 "per wave64 instruction, MI355X_MICROARCH.md).  Round 3 lowered the instructions per cell (0.97 -> %.2f) "
                                 "as well as the time: `value` is the figure to compare across rounds" % (
                                     round(valu_per_cell * 32 * wpl), 32 * wpl, wpl, tm_pipe["streams"], valu_per_cell),
                         "valu_per_cell": round(valu_per_cell, 4), "words_per_lane": wpl,
                         **roofline_from_profiles(valu_per_cell, launch_cells, wpl),
                         "frac_at_round2_instructions_per_cell": round(sustained_tops / valu_per_cell * (31.0 / 32.0) / VALU_PEAK_TOPS, 4),
                         "one_launch_alone": {"cells": launch_cells, "avg_launch_us": round(tm["fill_ms"] * 1e3, 1),
                                              "achieved": round(alone_tops, 2), "frac": round(alone_tops / VALU_PEAK_TOPS, 4),
                                              "what": "HIP events around ONE launch with nothing else in flight (one workgroup "
                                                      "per compute unit; the timed region keeps two such launches in "
                                                      "flight); rocprofv3: kernel_duration_source"},
                         "calibrated": {"peak_gcups": round(calibrated_peak_gcups, 1),
                                        "frac_alone": round(launch_cells / fill_s / 1e9 / calibrated_peak_gcups, 4),
                                        "frac_sustained": round(value / args.gpus / calibrated_peak_gcups, 4),
                                        "what": "ceiling with every instruction kind at the issue cost measured alone on this "
                                                "chip (profiles/r01_valu_microbench.txt): %.1f cycles per 64 cells" % bits_cycles_per_64_cells(wpl)}},
            "roofline_hbm": {"bound": "hbm", "achieved": round(lp * impl_bytes / fill_s / 1e9, 1), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(lp * impl_bytes / fill_s / 1e9 / HBM_PEAK_GBS, 4),
                             "bytes_per_launch": int(lp * impl_bytes),
                             "algorithmic_bytes_per_launch": int(ALG_BYTES_PER_CELL * launch_cells),
                             "what": "bytes the fill actually writes (per lane and block of 32 steps: its planes + the carries it put "
                                     "out; no direction planes) / launch duration.  SURVEY 8(d)'s algorithmic 0.25 B/cell is "
                                     "listed for reference only: the kernel does not move those bytes, so it is not priced "
                                     "against HBM"},
        }
        if args.gpus == 1 and not args.no_extra_legs:
            # a throughput caller sizes its batches to fill the chip: 4 x the step's batch = one 512-workgroup
            # fill launch per batch (what the timed region reaches by merging 4 passes of 128 pairs)
            big = tasks if args.mode != "weak" else tasks + config4_tasks((world + 1) * args.pairs, 3 * args.pairs, args.length)
            line["streaming"] = streaming_leg(csa_amd, big, batches=8)
            line["streaming"]["pairs_per_batch"] = len(big)
            line["streaming"]["vs_value"] = round(line["streaming"]["gcups"] / (value / args.gpus), 3)
            line["one_shot"] = one_shot_leg(csa_amd, tasks)
            line["one_shot"]["vs_value"] = round(line["one_shot"]["gcups_device"] / (value / args.gpus), 3)
            line["profile_path"] = profile_path_leg(csa_amd)
            if dropin:
                # the same sets through csadp_msa (the library's own caller: all gaps of a set in one batch) beside the drop-in's modes
                for name, rec in dropin.items():
                    pp = line["profile_path"].get(name) if isinstance(rec, dict) else None
                    if pp and rec.get("deferred"):
                        rec["csadp_msa_dp_ms"] = pp["dp_ms"]
                        rec["deferred_dp_vs_csadp_msa_dp"] = round(rec["deferred"]["dp_s"] * 1e3 / pp["dp_ms"], 2)
                line["dropin"] = dropin
            line["single_matrix"] = single_matrix_leg(csa_amd)
            line["profile_batch"] = profile_batch_leg(csa_amd)
            # SURVEY 8(d)'s unit of work counts H2D of the sequences and D2H of the results inside the wall time: that rate,
            # first class beside `value` (which starts and ends in HBM, as the bench contract asks)
            line["gcups_8d_h2d_d2h_inclusive"] = {"value": line["streaming"]["gcups"], "unit": "GCUPS", "vs_value": line["streaming"]["vs_value"],
                                                  "what": "SURVEY 8(d): host letters -> H2D -> kernels -> D2H -> host strings inside the wall "
                                                          "time (the `streaming` leg: %d batches of %d pairs, %d in flight)" % (
                                                              line["streaming"]["batches"], len(big), line["streaming"]["in_flight"])}
            if args.mode == "weak" and args.workload == "config4":
                line["real_sets"] = real_sets_leg(csa_amd)
                line["config5"] = config5_leg(csa_amd)
                line["unrelated_16k"] = unrelated_leg(csa_amd)
                line["small_pairs"] = small_pairs_leg(csa_amd)
                if args.pairs <= 1024 and args.length == 16384:
                    line["config4_all"] = config4_all_leg(csa_amd, tasks)
                line["records"]["checked_against_reference_digests_in_all_legs"] = (
                    (line["config4_all"]["equal_to_reference_digests"] if "config4_all" in line else checked) +
                    line["real_sets"]["Mammals"]["equal_to_reference_digests"] + line["real_sets"]["Primates"]["equal_to_reference_digests"] +
                    line["config5"]["equal_to_reference_digests"] + line["unrelated_16k"]["equal_to_reference_digests"])
        if args.gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(tasks, results)
            line["cpu_baseline"]["many_cores"] = many_cores
            line["cpu_baseline"]["many_cores_o3"] = many_cores_o3
            line["cpu_baseline"]["many_cores_config3"] = many_cores_c3
            line["cpu_baseline"]["host"] = host_description()
            line["cpu_baseline"]["reference_faithful"] = reference_faithful_sample(csa_amd)
    batch.close()
    group.barrier()
    group.close()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())


if __name__ == "__main__":
    main()
