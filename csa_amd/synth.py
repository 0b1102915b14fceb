"""Deterministic synthetic workloads -- numpy, counter-based.

Pair p draws from splitmix64 seeded 0x9E3779B97F4A7C15*(p+1).  Stream layout (index n of
the stream = n-th output of the generator):
    [0, L)                  bases of a:  "ACGT"[x & 3]
    L + 5*j + {0..4}        for base j of a:  u_del, u_ins, inserted base, u_sub, substitute offset
    L + 5*L                 rotation r = x % len(b)
u_* = (x >> 11) / 2^53.  b = a with per-base deletion (u_del < 1%), else optional insertion
before the base (u_ins < 1%), then substitution (u_sub < 10%, to one of the 3 other bases);
b is left-rotated by r and the task rotations {0, (len(b)-r) % len(b)} re-linearise it.
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def splitmix64_stream(seed, start, count):
    """Outputs start .. start+count-1 of splitmix64(seed) as uint64 (vectorised)."""
    with np.errstate(over="ignore"):
        n = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + n * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def synth_pair(p, length=16384, sub=0.10, ins=0.01, dele=0.01, unrelated=False):
    """Returns (a, b_rotated, rot_a, rot_b) as bytes/ints."""
    seed = (0x9E3779B97F4A7C15 * (p + 1)) & 0xFFFFFFFFFFFFFFFF
    L = int(length)
    x = splitmix64_stream(seed, 0, 6 * L + 1)
    a = _ACGT[(x[:L] & np.uint64(3)).astype(np.int64)]
    d = x[L:6 * L].reshape(L, 5)
    if unrelated:
        b = _ACGT[(d[:, 2] & np.uint64(3)).astype(np.int64)]
    else:
        u = (d >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        keep = u[:, 0] >= dele
        do_ins = keep & (u[:, 1] < ins)
        do_sub = u[:, 3] < sub
        acode = (x[:L] & np.uint64(3)).astype(np.int64)
        scode = (acode + 1 + (d[:, 4] % np.uint64(3)).astype(np.int64)) & 3
        base = np.where(do_sub, scode, acode)
        icode = (d[:, 2] & np.uint64(3)).astype(np.int64)
        # interleave: optional inserted base before each kept base
        out = np.empty(2 * L, dtype=np.int64)
        mask = np.zeros(2 * L, dtype=bool)
        out[0::2] = icode
        mask[0::2] = do_ins
        out[1::2] = base
        mask[1::2] = keep
        b = _ACGT[out[mask]]
    r = int(x[6 * L] % np.uint64(len(b)))
    brot = np.concatenate([b[r:], b[:r]])
    return a.tobytes(), brot.tobytes(), 0, (len(b) - r) % len(b)


def config4_tasks(first, count, length=16384):
    """Tasks [first, first+count) of the 1024-pair synthetic batch (config 4)."""
    tasks = []
    for p in range(first, first + count):
        a, b, ra, rb = synth_pair(p, length)
        tasks.append(([a, b], [ra, rb], None, None))
    return tasks


def config5_lengths(count=256, seed=5):
    """Mixed-length batch (config 5): L = round(10^u), u ~ U[3, 5.30103]; partner 0.95..1.05 L."""
    x = splitmix64_stream(seed, 0, 2 * count)
    u = (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    la = np.rint(10.0 ** (3.0 + u[:count] * 2.30103)).astype(np.int64)
    lb = np.rint(la * (0.95 + 0.10 * u[count:])).astype(np.int64)
    return la.tolist(), lb.tolist()
