"""csa_amd -- Python binding (ctypes) of libcsadp.so, the MI355X implementation of CSA's
dynamic-programming alignment hot path (reference: source/dynamicprogramming.c).

The binding mirrors the C-ABI in include/csadp.h one to one; it exists for the test-suite
and bench.py.  There is no Python or CPU implementation behind it: if the shared library is
missing or no gfx950 device is usable, calls raise CsadpError.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcsadp.so")

OK = 0
ERR_ARG, ERR_ALPHABET, ERR_NOMEM, ERR_NO_DEVICE, ERR_HIP, ERR_RANGE, ERR_STATE = -1, -2, -3, -4, -5, -6, -7


class CsadpError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = lib().csadp_strerror(code).decode() if _lib is not None else "library not loaded"
        super().__init__("%s: %s (%d)" % (what, msg, code) if what else "%s (%d)" % (msg, code))


class Config(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int), ("tile_rows", ctypes.c_int), ("verbose", ctypes.c_int)]


class Task(ctypes.Structure):
    _fields_ = [("nseq", ctypes.c_int),
                ("texts", ctypes.POINTER(ctypes.c_char_p)),
                ("textsizes", ctypes.POINTER(ctypes.c_int)),
                ("rotations", ctypes.POINTER(ctypes.c_int)),
                ("starts", ctypes.POINTER(ctypes.c_int)),
                ("ends", ctypes.POINTER(ctypes.c_int))]


class Result(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int), ("score", ctypes.c_int), ("consensus", ctypes.c_int),
                ("fills", ctypes.c_int), ("cells", ctypes.c_longlong),
                ("aligned", ctypes.POINTER(ctypes.c_void_p)), ("progress", ctypes.c_void_p)]


class SpStats(ctypes.Structure):
    _fields_ = [("consensus", ctypes.c_int), ("total_gaps", ctypes.c_longlong),
                ("conserved_columns", ctypes.c_int), ("sp_score", ctypes.c_longlong)]


class RotationInfo(ctypes.Structure):
    _fields_ = [("blocks", ctypes.c_int), ("chain_size", ctypes.c_int), ("chain_span", ctypes.c_int),
                ("first_block_depth", ctypes.c_int)]


class AnchorMap(ctypes.Structure):
    _fields_ = [("nseq", ctypes.c_int), ("nsegs", ctypes.c_int), ("border_nodes", ctypes.c_int),
                ("size", ctypes.POINTER(ctypes.c_int)), ("dp", ctypes.POINTER(ctypes.c_int)),
                ("positions", ctypes.POINTER(ctypes.c_int))]


class MsaStats(ctypes.Structure):
    _fields_ = [("nseq", ctypes.c_int), ("border_nodes", ctypes.c_int), ("segments", ctypes.c_int),
                ("dp_gaps", ctypes.c_int), ("fills", ctypes.c_int), ("alignment_length", ctypes.c_int),
                ("cells", ctypes.c_longlong), ("rotations_ms", ctypes.c_double), ("anchors_ms", ctypes.c_double),
                ("dp_ms", ctypes.c_double), ("rows_ms", ctypes.c_double), ("recoveries", ctypes.c_int)]


class Timing(ctypes.Structure):
    _fields_ = [("cells", ctypes.c_longlong), ("fill_launches", ctypes.c_int),
                ("fill_tiles", ctypes.c_longlong), ("fill_ms", ctypes.c_float),
                ("traceback_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
                ("dir_bytes", ctypes.c_longlong), ("border_bytes", ctypes.c_longlong),
                ("launch_passes", ctypes.c_int), ("bit_parallel", ctypes.c_int),
                ("merge_group", ctypes.c_int), ("recoveries", ctypes.c_int), ("device_io", ctypes.c_int),
                ("words_per_lane", ctypes.c_int), ("streams", ctypes.c_int)]


# symbols declared in include/csadp.h and include/csadp_debug.h
EXPORTS = [
    "csadp_init", "csadp_warmup", "csadp_shutdown", "csadp_version", "csadp_strerror", "csadp_device_info",
    "csadp_align_batch", "csadp_last_batch_phases", "csadp_recoveries", "csadp_free_result", "csadp_free_results", "csadp_device_count", "csadp_align_batch_on", "csadp_task_cost",
    "csadp_align_batch_multi", "csadp_pairs_create_on",
    "csadp_pairs_create", "csadp_pairs_run", "csadp_pairs_flush", "csadp_pairs_sync", "csadp_pairs_fetch",
    "csadp_pairs_destroy", "csadp_pairs_timing",
    "csadp_partition_lpt", "csadp_fnv1a", "csadp_load_fasta", "csadp_free_fasta",
    "csadp_sp_score", "csadp_write_rotated_fasta", "csadp_read_rotations", "csadp_score_pairs", "csadp_find_rotations",
    "csadp_build_anchor_map", "csadp_free_anchor_map", "csadp_msa", "csadp_free_rows", "csadp_write_aligned_fasta",
    "csadp_debug_align_with_filler", "csadp_debug_align_batch_with_filler", "csadp_debug_reload_config", "csadp_debug_pool_selftest", "csadp_debug_set_epoch",
]

DEBUG_FILL_FN = ctypes.CFUNCTYPE(
    ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
    ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_byte), ctypes.POINTER(ctypes.c_int), ctypes.c_int,
    ctypes.POINTER(ctypes.c_ubyte), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
    ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int))

_lib = None


def lib():
    """Load libcsadp.so (built by __graft_entry__.build() / make -C csa_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("csa_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'`; "
                               "there is no fallback implementation" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.csadp_strerror.restype = ctypes.c_char_p
        L.csadp_strerror.argtypes = [ctypes.c_int]
        L.csadp_init.argtypes = [ctypes.POINTER(Config)]
        L.csadp_align_batch.argtypes = [ctypes.POINTER(Task), ctypes.c_int, ctypes.POINTER(Result)]
        L.csadp_free_result.argtypes = [ctypes.POINTER(Result), ctypes.c_int]
        L.csadp_free_results.argtypes = [ctypes.POINTER(Result), ctypes.c_int, ctypes.c_int]
        L.csadp_pairs_create.argtypes = [ctypes.POINTER(Task), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        for name in ("csadp_pairs_run", "csadp_pairs_sync", "csadp_pairs_flush"):
            getattr(L, name).argtypes = [ctypes.c_void_p]
        L.csadp_pairs_destroy.argtypes = [ctypes.c_void_p]
        L.csadp_pairs_destroy.restype = None
        L.csadp_pairs_fetch.argtypes = [ctypes.c_void_p, ctypes.POINTER(Result)]
        L.csadp_pairs_timing.argtypes = [ctypes.c_void_p, ctypes.POINTER(Timing)]
        L.csadp_partition_lpt.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int, ctypes.c_int,
                                          ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_longlong)]
        L.csadp_debug_align_with_filler.argtypes = [ctypes.POINTER(Task), DEBUG_FILL_FN, ctypes.c_void_p,
                                                    ctypes.POINTER(Result)]
        _lib = L
    return _lib


def _check(code, what=""):
    if code != OK:
        raise CsadpError(code, what)


def init(device=-1, tile_rows=0, verbose=0):
    cfg = Config(device, tile_rows, verbose)
    _check(lib().csadp_init(ctypes.byref(cfg)), "csadp_init")


def warmup():
    _check(lib().csadp_warmup(), "csadp_warmup")


def shutdown():
    lib().csadp_shutdown()


def reload_config():
    """The library reads its environment switches once per process; the test-suite flips them between calls: read them again."""
    lib().csadp_debug_reload_config()


def device_info():
    name = ctypes.create_string_buffer(256)
    cus = ctypes.c_int()
    _check(lib().csadp_device_info(name, 256, ctypes.byref(cus)), "csadp_device_info")
    return name.value.decode(), cus.value


class TaskArray:
    """Owns the ctypes memory of an array of csadp_task (texts are borrowed by the library
    until the batch is fetched, so this object must stay alive that long)."""

    def __init__(self, tasks):
        # tasks: iterable of (texts, rotations, starts, ends); starts/ends may be None
        self.n = len(tasks)
        self.arr = (Task * self.n)()
        self._keep = []
        self.nseq = []
        for t, (texts, rots, starts, ends) in enumerate(tasks):
            n = len(texts)
            bts = [x if isinstance(x, bytes) else x.encode() for x in texts]
            rots = list(rots) if rots is not None else [0] * n
            starts = list(starts) if starts is not None else [0] * n
            ends = list(ends) if ends is not None else [len(b) for b in bts]
            c_txt = (ctypes.c_char_p * n)(*bts)
            c_sz = (ctypes.c_int * n)(*[len(b) for b in bts])
            c_rt = (ctypes.c_int * n)(*rots)
            c_st = (ctypes.c_int * n)(*starts)
            c_en = (ctypes.c_int * n)(*ends)
            self._keep.append((bts, c_txt, c_sz, c_rt, c_st, c_en))
            self.arr[t] = Task(n, c_txt, c_sz, c_rt, c_st, c_en)
            self.nseq.append(n)


def _unpack(results, nseqs):
    out = []
    L = lib()
    for r, n in zip(results, nseqs):
        strs = None
        if r.status == OK and r.aligned:
            strs = [ctypes.string_at(r.aligned[i]) for i in range(n)]
        out.append({"status": r.status, "score": r.score, "consensus": r.consensus, "fills": r.fills,
                    "cells": r.cells, "aligned": strs,
                    "progress": ctypes.string_at(r.progress).decode() if r.progress else ""})
        L.csadp_free_result(ctypes.byref(r), n)
    return out


def align_batch(tasks):
    """csadp_align_batch: tasks = [(texts, rotations, starts, ends), ...] -> list of dicts."""
    ta = TaskArray(tasks)
    res = (Result * max(ta.n, 1))()
    _check(lib().csadp_align_batch(ta.arr, ta.n, res), "csadp_align_batch")
    return _unpack(res[:ta.n], ta.nseq)


class BatchPhases(ctypes.Structure):
    _fields_ = [("tasks", ctypes.c_int), ("rounds", ctypes.c_int), ("round_groups", ctypes.c_int)] + [
        (k, ctypes.c_double) for k in ("wall_ms", "seed_ms", "layout_ms", "tables_ms", "device_ms", "apply_ms", "refine_speculate_ms",
                                       "refine_commit_ms", "results_ms")]


def recoveries():
    """Passes the primary engine's batches have repeated chunk by chunk so far in this process (0 in any healthy run)."""
    L = lib()
    L.csadp_recoveries.restype = ctypes.c_long
    return int(L.csadp_recoveries())


def last_batch_phases():
    """csadp_last_batch_phases: where the time of the last csadp_align_batch went (host clocks, summed over rounds and round groups)."""
    ph = BatchPhases()
    _check(lib().csadp_last_batch_phases(ctypes.byref(ph)), "csadp_last_batch_phases")
    return {k: getattr(ph, k) for k, _ in BatchPhases._fields_}


class MultiStats(ctypes.Structure):
    _fields_ = [("ndevices", ctypes.c_int), ("tasks", ctypes.c_int * 16), ("cost", ctypes.c_longlong * 16),
                ("ms", ctypes.c_double * 16), ("total_cost", ctypes.c_longlong), ("max_cost", ctypes.c_longlong),
                ("wall_ms", ctypes.c_double)]


def device_count():
    n = ctypes.c_int()
    rc = lib().csadp_device_count(ctypes.byref(n))
    return n.value if rc == OK else 0


def task_cost(task):
    ta = TaskArray([task])
    L = lib()
    L.csadp_task_cost.restype = ctypes.c_longlong
    L.csadp_task_cost.argtypes = [ctypes.POINTER(Task)]
    return L.csadp_task_cost(ta.arr)


def align_batch_on(device, tasks):
    """csadp_align_batch_on: the batch on an explicitly named HIP device."""
    ta = TaskArray(tasks)
    res = (Result * max(ta.n, 1))()
    L = lib()
    L.csadp_align_batch_on.argtypes = [ctypes.c_int, ctypes.POINTER(Task), ctypes.c_int, ctypes.POINTER(Result)]
    _check(L.csadp_align_batch_on(device, ta.arr, ta.n, res), "csadp_align_batch_on")
    return _unpack(res[:ta.n], ta.nseq)


def align_batch_multi(tasks, devices):
    """csadp_align_batch_multi: one batch over several GPUs of this node from one process
    (LPT partition, one host thread per GPU).  devices = list of HIP ordinals.  Returns (results, stats)."""
    ta = TaskArray(tasks)
    res = (Result * max(ta.n, 1))()
    st = MultiStats()
    dev = (ctypes.c_int * len(devices))(*devices)
    L = lib()
    L.csadp_align_batch_multi.argtypes = [ctypes.POINTER(Task), ctypes.c_int, ctypes.POINTER(Result), ctypes.POINTER(ctypes.c_int),
                                          ctypes.c_int, ctypes.POINTER(MultiStats)]
    _check(L.csadp_align_batch_multi(ta.arr, ta.n, res, dev, len(devices), ctypes.byref(st)), "csadp_align_batch_multi")
    n = st.ndevices
    stats = {"ndevices": n, "tasks": list(st.tasks[:n]), "cost": list(st.cost[:n]), "ms": list(st.ms[:n]),
             "total_cost": st.total_cost, "max_cost": st.max_cost, "wall_ms": st.wall_ms}
    return _unpack(res[:ta.n], ta.nseq), stats


class PairBatch:
    """Device-resident batch of 2-sequence tasks (csadp_pairs_*)."""

    def __init__(self, tasks, device=None):
        self.ta = TaskArray(tasks)
        self.h = ctypes.c_void_p()
        if device is None:
            _check(lib().csadp_pairs_create(self.ta.arr, self.ta.n, ctypes.byref(self.h)), "csadp_pairs_create")
        else:
            L = lib()
            L.csadp_pairs_create_on.argtypes = [ctypes.c_int, ctypes.POINTER(Task), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
            _check(L.csadp_pairs_create_on(device, self.ta.arr, self.ta.n, ctypes.byref(self.h)), "csadp_pairs_create_on")

    def run(self):
        _check(lib().csadp_pairs_run(self.h), "csadp_pairs_run")

    def sync(self):
        _check(lib().csadp_pairs_sync(self.h), "csadp_pairs_sync")

    def flush(self):
        _check(lib().csadp_pairs_flush(self.h), "csadp_pairs_flush")

    def timing(self):
        t = Timing()
        _check(lib().csadp_pairs_timing(self.h, ctypes.byref(t)), "csadp_pairs_timing")
        return {k: getattr(t, k) for k, _ in Timing._fields_}

    def fetch(self):
        res = (Result * self.ta.n)()
        _check(lib().csadp_pairs_fetch(self.h, res), "csadp_pairs_fetch")
        return _unpack(res[:self.ta.n], self.ta.nseq)

    def close(self):
        if self.h:
            lib().csadp_pairs_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def score_pairs(tasks):
    """csadp_score_pairs: DP scores of 2-sequence tasks without building strings."""
    ta = TaskArray(tasks)
    scores = (ctypes.c_int * ta.n)()
    status = (ctypes.c_int * ta.n)()
    L = lib()
    L.csadp_score_pairs.argtypes = [ctypes.POINTER(Task), ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    _check(L.csadp_score_pairs(ta.arr, ta.n, scores, status), "csadp_score_pairs")
    return list(scores), list(status)


def find_rotations(texts):
    """csadp_find_rotations: rotation offsets as the reference's tree analysis picks them.
    Returns (status, rotations, info)."""
    n = len(texts)
    bts = [t if isinstance(t, bytes) else t.encode() for t in texts]
    arr = (ctypes.c_char_p * n)(*bts)
    sz = (ctypes.c_int * n)(*[len(b) for b in bts])
    rot = (ctypes.c_int * n)()
    info = RotationInfo()
    L = lib()
    L.csadp_find_rotations.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int),
                                       ctypes.POINTER(ctypes.c_int), ctypes.POINTER(RotationInfo)]
    rc = L.csadp_find_rotations(n, arr, sz, rot, ctypes.byref(info))
    return rc, list(rot), {k: getattr(info, k) for k, _ in RotationInfo._fields_}


def build_anchor_map(texts, rotations):
    """csadp_build_anchor_map: the reference's alignment map for the given rotations.
    Returns (status, segments, border_nodes); segments = [(size, dp, [positions])]."""
    n = len(texts)
    bts = [t if isinstance(t, bytes) else t.encode() for t in texts]
    arr = (ctypes.c_char_p * n)(*bts)
    sz = (ctypes.c_int * n)(*[len(b) for b in bts])
    rot = (ctypes.c_int * n)(*rotations)
    m = AnchorMap()
    L = lib()
    L.csadp_build_anchor_map.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int),
                                         ctypes.POINTER(ctypes.c_int), ctypes.POINTER(AnchorMap)]
    rc = L.csadp_build_anchor_map(n, arr, sz, rot, ctypes.byref(m))
    segs = []
    if rc == 0:
        for k in range(m.nsegs):
            segs.append((m.size[k], m.dp[k], [m.positions[k * n + s] for s in range(n)]))
    nodes = m.border_nodes
    L.csadp_free_anchor_map.argtypes = [ctypes.POINTER(AnchorMap)]
    L.csadp_free_anchor_map.restype = None
    L.csadp_free_anchor_map(ctypes.byref(m))
    return rc, segs, nodes


def msa(texts, rotations=None):
    """csadp_msa: the reference's alignment stage (mode N when rotations is None, else the given
    rotations).  Returns (status, rotations, rows, stats)."""
    n = len(texts)
    bts = [t if isinstance(t, bytes) else t.encode() for t in texts]
    arr = (ctypes.c_char_p * n)(*bts)
    sz = (ctypes.c_int * n)(*[len(b) for b in bts])
    rin = (ctypes.c_int * n)(*rotations) if rotations is not None else None
    rout = (ctypes.c_int * n)()
    rows = ctypes.POINTER(ctypes.c_char_p)()
    st = MsaStats()
    L = lib()
    L.csadp_msa.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int),
                            ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                            ctypes.POINTER(ctypes.POINTER(ctypes.c_char_p)), ctypes.POINTER(MsaStats)]
    rc = L.csadp_msa(n, arr, sz, rin, rout, ctypes.byref(rows), ctypes.byref(st))
    out = []
    if rc == 0:
        out = [rows[s] for s in range(n)]
        L.csadp_free_rows.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int]
        L.csadp_free_rows.restype = None
        L.csadp_free_rows(rows, n)
    return rc, list(rout), out, {k: getattr(st, k) for k, _ in MsaStats._fields_}


def write_aligned_fasta(path, descs, rotations, rows):
    n = len(rows)
    d = (ctypes.c_char_p * n)(*[x if isinstance(x, bytes) else x.encode() for x in descs])
    r = (ctypes.c_char_p * n)(*[x if isinstance(x, bytes) else x.encode() for x in rows])
    rot = (ctypes.c_int * n)(*rotations) if rotations is not None else None
    L = lib()
    L.csadp_write_aligned_fasta.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int),
                                            ctypes.POINTER(ctypes.c_char_p), ctypes.c_int]
    _check(L.csadp_write_aligned_fasta(path.encode(), d, rot, r, n), "csadp_write_aligned_fasta")


def fnv1a(strs):
    """csadp_fnv1a over a list of byte strings."""
    n = len(strs)
    arr = (ctypes.c_char_p * n)(*strs)
    L = lib()
    L.csadp_fnv1a.restype = ctypes.c_uint
    L.csadp_fnv1a.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int]
    return L.csadp_fnv1a(arr, n)


def partition_lpt(costs, nparts):
    n = len(costs)
    c = (ctypes.c_longlong * max(n, 1))(*costs)
    a = (ctypes.c_int * max(n, 1))()
    m = ctypes.c_longlong()
    _check(lib().csadp_partition_lpt(c, n, nparts, a, ctypes.byref(m)), "csadp_partition_lpt")
    return list(a[:n]), m.value


def load_fasta(path):
    texts = ctypes.POINTER(ctypes.c_char_p)()
    descs = ctypes.POINTER(ctypes.c_char_p)()
    sizes = ctypes.POINTER(ctypes.c_int)()
    n = ctypes.c_int()
    L = lib()
    _check(L.csadp_load_fasta(path.encode(), ctypes.byref(texts), ctypes.byref(descs), ctypes.byref(sizes),
                              ctypes.byref(n)), "csadp_load_fasta")
    out = [(descs[i].decode(errors="replace"), bytes(texts[i])) for i in range(n.value)]
    L.csadp_free_fasta(texts, descs, sizes, n.value)
    return out


def sp_score(aligned):
    """csadp_sp_score: column statistics of aligned strings on the GPU (tools.c:194-293)."""
    n = len(aligned)
    arr = (ctypes.c_char_p * n)(*aligned)
    st = SpStats()
    L = lib()
    L.csadp_sp_score.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.POINTER(SpStats)]
    _check(L.csadp_sp_score(arr, n, ctypes.byref(st)), "csadp_sp_score")
    return {k: getattr(st, k) for k, _ in SpStats._fields_}


def write_rotated_fasta(path, records, rotations):
    """records = [(desc, text_bytes)], csadp_write_rotated_fasta (csamsa.c:416-431)."""
    n = len(records)
    descs = (ctypes.c_char_p * n)(*[d.encode() for d, _ in records])
    texts = (ctypes.c_char_p * n)(*[t for _, t in records])
    sizes = (ctypes.c_int * n)(*[len(t) for _, t in records])
    rots = (ctypes.c_int * n)(*rotations)
    _check(lib().csadp_write_rotated_fasta(path.encode(), descs, texts, sizes, rots, n), "csadp_write_rotated_fasta")


def read_rotations(path, nmax=64):
    rots = (ctypes.c_int * nmax)()
    n = ctypes.c_int()
    _check(lib().csadp_read_rotations(path.encode(), rots, nmax, ctypes.byref(n)), "csadp_read_rotations")
    return list(rots[:n.value])


def debug_set_epoch(value):
    """Test seam: set the process-wide epoch counter of the chunked fills' hand-off granules; returns the old value."""
    L = lib()
    L.csadp_debug_set_epoch.argtypes = [ctypes.c_uint]
    L.csadp_debug_set_epoch.restype = ctypes.c_uint
    return L.csadp_debug_set_epoch(value)


def debug_align_with_filler(task, filler):
    """Host-logic test seam (include/csadp_debug.h): `filler` is a DEBUG_FILL_FN-compatible
    Python callable supplied by the test-suite."""
    ta = TaskArray([task])
    res = Result()
    cb = DEBUG_FILL_FN(filler)
    rc = lib().csadp_debug_align_with_filler(ta.arr, cb, None, ctypes.byref(res))
    if rc != OK:
        return {"status": rc, "aligned": None, "consensus": 0, "score": 0, "fills": 0, "cells": 0}
    return _unpack([res], ta.nseq)[0]


def debug_align_batch_with_filler(tasks, filler):
    """The same seam for a batch: the product's round driver (lock-step rounds, round groups on their own host threads,
    CSADP_ROUND_GROUPS) with `filler` in place of the device step.  The filler is called from several threads."""
    ta = TaskArray(tasks)
    res = (Result * ta.n)()
    cb = DEBUG_FILL_FN(filler)
    L = lib()
    L.csadp_debug_align_batch_with_filler.argtypes = [ctypes.POINTER(Task), ctypes.c_int, DEBUG_FILL_FN, ctypes.c_void_p, ctypes.POINTER(Result)]
    _check(L.csadp_debug_align_batch_with_filler(ta.arr, ta.n, cb, None, res), "csadp_debug_align_batch_with_filler")
    return _unpack(res[:ta.n], ta.nseq)
