"""Shared test helpers: ctypes bindings for the oracle / reference checker
libraries, FASTA parsing (rules of /root/reference/source/csamsa.c:482-490),
deterministic synthetic sequence generators.

Everything under oracle/ is TEST INFRASTRUCTURE: it is loaded here as the
checker only, never by the product package (csa_amd).
"""
import ctypes
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libcsa_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libcsa_ref.so")
REF_O3_SO = os.path.join(ROOT, "oracle", "_ref", "libcsa_ref_o3.so")     # the same reference sources at -O3 -march=znver3 (oracle/Makefile)
GOLDEN = os.path.join(ROOT, "tests", "golden")


class OdpStats(ctypes.Structure):
    _fields_ = [("cells", ctypes.c_longlong), ("fills", ctypes.c_int),
                ("stale_border_fills", ctypes.c_int), ("last_score", ctypes.c_int),
                ("consensus", ctypes.c_int), ("fill_seconds", ctypes.c_double)]


_oracle = None
_ref = {}


def build_oracle():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        lib = ctypes.CDLL(ORACLE_SO)
        lib.odp_progressive_dp.restype = ctypes.c_int
        lib.odp_sp_score.restype = ctypes.c_longlong
        lib.odp_fnv1a.restype = ctypes.c_uint
        lib.odp_fill.restype = ctypes.c_int
        _oracle = lib
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def have_ref_o3():
    return os.path.exists(REF_O3_SO)


def ref_lib(o3=False):
    path = REF_O3_SO if o3 else REF_SO
    if path not in _ref:
        lib = ctypes.CDLL(path)
        lib.csa_ref_progressive_dp.restype = ctypes.c_int
        _ref[path] = lib
    return _ref[path]


def _task_args(texts, rots, starts, ends):
    n = len(texts)
    bts = [t if isinstance(t, bytes) else t.encode() for t in texts]
    txt = (ctypes.c_char_p * n)(*bts)
    sz = (ctypes.c_int * n)(*[len(t) for t in bts])
    rt = (ctypes.c_int * n)(*rots)
    st = (ctypes.c_int * n)(*starts)
    en = (ctypes.c_int * n)(*ends)
    return n, txt, sz, rt, st, en


def _collect(out, n, free):
    strs = []
    for i in range(n):
        if out[i]:
            strs.append(ctypes.string_at(out[i]))
            free(ctypes.c_void_p(out[i]))
        else:
            strs.append(None)
    return strs


def oracle_progressive(texts, rots=None, starts=None, ends=None):
    """Run the CPU restatement.  Returns (consensus_or_error, strings, stats)."""
    lib = oracle_lib()
    n = len(texts)
    rots = rots or [0] * n
    starts = starts or [0] * n
    ends = ends or [len(t) for t in texts]
    n, txt, sz, rt, st, en = _task_args(texts, rots, starts, ends)
    out = (ctypes.c_void_p * n)()
    stats = OdpStats()
    rc = lib.odp_progressive_dp(n, txt, sz, rt, st, en, out, ctypes.byref(stats))
    strs = _collect(out, n, lib.odp_free) if rc >= 0 else [None] * n
    return rc, strs, stats


def ref_progressive(texts, rots=None, starts=None, ends=None, o3=False):
    """Run the compiled reference (oracle/_ref; o3: its -O3 build).  Returns (consensus, strings, seconds)."""
    lib = ref_lib(o3)
    n = len(texts)
    rots = rots or [0] * n
    starts = starts or [0] * n
    ends = ends or [len(t) for t in texts]
    n, txt, sz, rt, st, en = _task_args(texts, rots, starts, ends)
    out = (ctypes.c_void_p * n)()
    sec = ctypes.c_double()
    rc = lib.csa_ref_progressive_dp(n, txt, sz, rt, st, en, out, ctypes.byref(sec))
    strs = _collect(out, n, lib.csa_ref_free)
    return rc, strs, sec.value


def oracle_pair_score_linear(texts, rots=None, starts=None, ends=None):
    """dpmatrix[nrows][ncols] of a 2-sequence task from the oracle's two-row (linear-space) fill:
    the optimality check for pairs too long for the reference's 5 B/cell matrices."""
    lib = oracle_lib()
    rots = rots or [0, 0]
    starts = starts or [0, 0]
    ends = ends or [len(t) for t in texts]
    n, txt, sz, rt, st, en = _task_args(texts, rots, starts, ends)
    assert n == 2
    score = ctypes.c_longlong()
    rc = lib.odp_pair_score_linear(txt, sz, rt, st, en, ctypes.byref(score))
    assert rc == 0, rc
    return score.value


def oracle_sp_stats(strs):
    """The reference's mode-S statistics (tools.c:194-293) from the oracle's restatement, shaped
    like tests/golden/*'s "mode_s" records."""
    lib = oracle_lib()
    n = len(strs)
    arr = (ctypes.c_char_p * n)(*[s if isinstance(s, bytes) else s.encode() for s in strs])
    cons, conserved = ctypes.c_int(), ctypes.c_int()
    gaps, sp = ctypes.c_longlong(), ctypes.c_longlong()
    rc = lib.odp_sp_stats(n, arr, ctypes.byref(cons), ctypes.byref(gaps), ctypes.byref(conserved), ctypes.byref(sp))
    assert rc == 0, rc
    return {"consensus": cons.value, "avg_gaps": gaps.value // n, "conserved": conserved.value, "sp": sp.value}


def sp_score(strs):
    """Sum-of-pairs score, rule of tools.c:274-280 (numpy restatement)."""
    arrs = [np.frombuffer(s, dtype=np.uint8) for s in strs]
    score = 0
    for i in range(len(arrs)):
        for j in range(i + 1, len(arrs)):
            both = (arrs[i] == 45) & (arrs[j] == 45)
            eq = (arrs[i] == arrs[j]) & ~both
            ne = arrs[i] != arrs[j]
            score += int(eq.sum()) - int(ne.sum())
    return score


def fnv1a(strs):
    h = 0x811C9DC5
    for s in strs:
        for ch in s:
            h ^= ch
            h = (h * 0x01000193) & 0xFFFFFFFF
    return h


def degap(s):
    return s.replace(b"-", b"")


def rotated(text, rot, start=0, end=None):
    t = text if isinstance(text, bytes) else text.encode()
    end = len(t) if end is None else end
    r = t[rot:] + t[:rot]
    return r[start:end]


def read_fasta(path):
    """FASTA reader following csamsa.c:433-519 for well-formed ACGT input."""
    seqs, descs, cur = [], [], None
    with open(path, "rb") as f:
        for line in f:
            line = line.strip()
            if line.startswith(b">"):
                descs.append(line[1:].decode(errors="replace"))
                cur = []
                seqs.append(cur)
            elif cur is not None:
                cur.append(line.replace(b" ", b"").replace(b"-", b"").upper())
    return descs, [b"".join(s) for s in seqs]


# ---- deterministic synthetic inputs -----------------

sys.path.insert(0, ROOT)
from csa_amd.synth import synth_pair  # noqa: E402,F401  (host-side workload generator, numpy)


def random_family(rng, nseq, length, mut=0.15, indel=0.08, alphabet=b"ACGT"):
    """nseq related sequences for progressive-DP tests (python RNG, small sizes)."""
    base = bytes(rng.choice(alphabet) for _ in range(length))
    fam = []
    for _ in range(nseq):
        out = bytearray()
        for ch in base:
            x = rng.random()
            if x < indel / 2:
                continue
            if x < indel:
                out.append(rng.choice(alphabet))
            if rng.random() < mut:
                ch = rng.choice(alphabet)
            out.append(ch)
        fam.append(bytes(out))
    return fam


def rng(seed):
    return random.Random(seed)


# ---- oracle-backed matrix filler for the host-logic test seam (csadp_debug.h) -----------

def oracle_filler():
    """Returns a Python callable with the csadp_debug_fill_fn signature that fills the
    matrix with the oracle's odp_fill and walks the directions (dynamicprogramming.c
    :1037-1047).  Used ONLY by CPU tests of the product's host logic."""
    lib = oracle_lib()

    def fill(user, nrows, ncols, nprev, sv, rowcodes, top, left_i, ops, nops, remj, remk, score):
        H = (ctypes.c_int * ((nrows + 1) * (ncols + 1)))()
        D = ctypes.create_string_buffer((nrows + 1) * (ncols + 1))
        rc = lib.odp_fill(nrows, ncols, rowcodes, sv, nprev, top, left_i, H, D)
        if rc != 0:
            return -5
        j, k, n = nrows, ncols, 0
        pitch = ncols + 1
        raw = D.raw
        while j > 0 and k > 0:
            d = raw[j * pitch + k]
            if d == 68:      # 'D'
                ops[n] = 2
                j -= 1
                k -= 1
            elif d == 76:    # 'L'
                ops[n] = 1
                k -= 1
            else:            # 'U'
                ops[n] = 0
                j -= 1
            n += 1
        nops[0] = n
        remj[0] = j
        remk[0] = k
        score[0] = H[nrows * pitch + ncols]
        return 0

    return fill


def load_golden(name):
    import json
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def golden_task(case):
    return ([t.encode() for t in case["texts"]], case["rots"], case["starts"], case["ends"])


def golden_aligned(case):
    return [a.encode() if a is not None else None for a in case["aligned"]]


# ---- rotation finder checkers -------------------------------------------------------------

def ref_rotations(seqs, timeout=20):
    """Reference tree analysis (oracle/_ref, ref_shim.c:csa_ref_rotations) in a forked child: the
    reference calls exit()/getchar() when it finds no block and can loop forever on some inputs.
    Returns (rc, rotations, blocks); rc -9 = killed (non-terminating), -8 = exited."""
    import pickle
    import signal
    lib = ref_lib()
    n = len(seqs)
    rd, wr = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(rd)
        signal.alarm(timeout)
        devnull = os.open(os.devnull, os.O_RDONLY)
        os.dup2(devnull, 0)
        txt = (ctypes.c_char_p * n)(*seqs)
        sz = (ctypes.c_int * n)(*[len(s) for s in seqs])
        rot = (ctypes.c_int * n)()
        cap = 1 << 20
        dump = (ctypes.c_int * cap)()
        nb = ctypes.c_int()
        rc = lib.csa_ref_rotations(n, txt, sz, rot, dump, cap, ctypes.byref(nb))
        w = 4 + n
        blocks = [tuple(dump[i * w:i * w + 3]) + (list(dump[i * w + 4:(i + 1) * w]),) for i in range(min(nb.value, cap // w))]
        os.write(wr, pickle.dumps((rc, list(rot), blocks)))
        os._exit(0)
    os.close(wr)
    data = b""
    while True:
        chunk = os.read(rd, 1 << 16)
        if not chunk:
            break
        data += chunk
    os.close(rd)
    _, st = os.waitpid(pid, 0)
    if not data:
        return (-9 if os.WIFSIGNALED(st) else -8), [], []
    return pickle.loads(data)


def rotated_family(r, nseq, length, mut=0.05, indel=0.02):
    """Related circular sequences, each cut at a random point (inputs of the rotation finder)."""
    out = []
    for f in random_family(r, nseq, length, mut=mut, indel=indel):
        if len(f) < 12:
            f = f + b"ACGTTGCAAGCT"
        k = r.randrange(len(f))
        out.append(f[k:] + f[:k])
    return out


# ---- anchor map checkers -------------------------------------------------------------------

def ref_alignment_map(seqs, given_rot=None, savepath=None, timeout=60):
    """Reference N-mode alignment stage (oracle/_ref, ref_shim.c:csa_ref_alignment_map) in a forked
    child (the reference may exit() or not terminate).  Returns (rc, rotations, border, segments):
    border = [(size, [positions per sequence])], segments = [(size, has_dp, [positions])]."""
    import pickle
    import signal
    lib = ref_lib()
    n = len(seqs)
    rd, wr = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(rd)
        signal.alarm(timeout)
        devnull = os.open(os.devnull, os.O_RDONLY)
        os.dup2(devnull, 0)
        txt = (ctypes.c_char_p * n)(*seqs)
        sz = (ctypes.c_int * n)(*[len(s) for s in seqs])
        rot = (ctypes.c_int * n)()
        grot = (ctypes.c_int * n)(*given_rot) if given_rot is not None else None
        bcap = scap = 1 << 22
        bd = (ctypes.c_int * bcap)()
        sd = (ctypes.c_int * scap)()
        nb = ctypes.c_int()
        ns = ctypes.c_int()
        rc = lib.csa_ref_alignment_map(n, txt, sz, grot, rot, bd, bcap, ctypes.byref(nb), sd, scap, ctypes.byref(ns),
                                       savepath.encode() if savepath else None)
        border, at = [], 0
        for _ in range(nb.value if rc == 0 else 0):
            size = bd[at]
            at += 1
            pos = []
            for _s in range(n):
                c = bd[at]
                pos.append(list(bd[at + 1:at + 1 + c]))
                at += 1 + c
            border.append((size, pos))
        w = 2 + n
        segs = [(sd[i * w], sd[i * w + 1], list(sd[i * w + 2:(i + 1) * w])) for i in range(ns.value if rc == 0 else 0)]
        os.write(wr, pickle.dumps((rc, list(rot), border, segs)))
        os._exit(0)
    os.close(wr)
    data = b""
    while True:
        chunk = os.read(rd, 1 << 16)
        if not chunk:
            break
        data += chunk
    os.close(rd)
    _, st = os.waitpid(pid, 0)
    if not data:
        return (-9 if os.WIFSIGNALED(st) else -8), [], [], []
    return pickle.loads(data)
