"""Which shapes does the bit-parallel path get wrong?  For W in 1, 2, 4: pairs of many (rows, cols) shapes against the oracle;
prints every failing shape with the position of the first differing alignment column (counted from the right end)."""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import csa_amd
from helpers import oracle_progressive

csa_amd.init(device=0)
r = random.Random(5)

def related(n, m):
    a = bytes(r.choice(b"ACGT") for _ in range(n))
    out = bytearray()
    for ch in a:
        x = r.random()
        if x < 0.02:
            continue
        if x < 0.04:
            out.append(r.choice(b"ACGT"))
        out.append(r.choice(b"ACGT") if r.random() < 0.1 else ch)
    b = bytes(out)
    b = (b + bytes(r.choice(b"ACGT") for _ in range(m)))[:m]
    return a, b

shapes = []
for cols in (100, 1000, 2049, 4097, 5000, 8193, 9000):
    for rows in (cols, cols + 1, cols + 31, cols + 33, cols + 40, cols + 64, cols + 95, cols + 200, cols + 1000):
        shapes.append((rows, cols))
tasks = [([*related(rows, cols)], None, None, None) for rows, cols in shapes]
want = [oracle_progressive(t[0], t[1]) for t in tasks]
for W in (sys.argv[1:] or ["1", "2", "4"]):
    os.environ["CSADP_BITS_WORDS"] = W
    got = csa_amd.align_batch(tasks)
    bad = 0
    for (rows, cols), g, (cons, strs, st) in zip(shapes, got, want):
        if g["aligned"] != strs or g["score"] != st.last_score:
            bad += 1
            a, b = g["aligned"][0], strs[0]
            pos = next((i for i in range(1, min(len(a), len(b)) + 1) if a[-i] != b[-i]), -1)
            pos1 = next((i for i in range(1, min(len(g["aligned"][1]), len(strs[1])) + 1) if g["aligned"][1][-i] != strs[1][-i]), -1)
            print(f"W={W} rows={rows} cols={cols}: status {g['status']} consensus {g['consensus']} vs {cons}, score {g['score']} vs {st.last_score}, first difference {pos}/{pos1} columns from the right end")
    print(f"W={W}: {bad} of {len(shapes)} shapes differ")
