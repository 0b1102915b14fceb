"""oracle/rot_oracle.py -- TEST INFRASTRUCTURE ONLY.

Brute-force restatement of the reference's rotation finder for SMALL inputs, written from the
definitions (no suffix tree): which substrings become blocks (csamsa.c:69-109 collectNodes /
removeSuffixNodes, :230-257 removeNonUniqueNodes), how blocks are ordered and linked into
chains (:132-226 collectNodeChains, nodeslinkedlists.c:34-79), and which chain decides the
rotations (:260-267).  It exists to pin those semantics against oracle/_ref (ref_shim.c:
csa_ref_rotations) before the product implements them with an index structure.

Parity status: see tests/test_rotations.py.
"""


def cyc(seq, p, length):
    n = len(seq)
    if p + length <= n:
        return seq[p:p + length]
    return seq[p:] + seq[:p + length - n]


def occurrences(seq, w):
    n = len(seq)
    if len(w) > n:
        return []
    dbl = seq + seq[:len(w) - 1] if len(w) > 1 else seq
    out = []
    start = 0
    while True:
        i = dbl.find(w, start)
        if i < 0 or i >= n:
            break
        out.append(i)
        start = i + 1
    return out


def common_strings(seqs):
    """All strings that occur (cyclically) in every sequence."""
    minlen = min(len(s) for s in seqs)
    s0 = seqs[0]
    common = set()
    for p in range(len(s0)):
        for length in range(1, minlen + 1):
            w = cyc(s0, p, length)
            if w in common:
                continue
            if all(occurrences(s, w) for s in seqs[1:]):
                common.add(w)
            else:
                break          # longer prefixes of this suffix cannot be common either
    return common


def first_end(seqs, w):
    """(sequence index, end position in the doubled scan) at which w is first spelled while the
    sequences are inserted one after the other, each traversed twice (gencycsuffixtrees.c:426)."""
    for j, s in enumerate(seqs):
        n = len(s)
        dbl = s + s
        i = dbl.find(w)
        if 0 <= i and len(w) <= n:
            return (j, i + len(w))
    return (len(seqs), 0)


def find_blocks(seqs):
    common = common_strings(seqs)
    collected = [w for w in common if not any((w + c) in common for c in (b"A", b"C", b"G", b"T"))]
    cset = set(collected)
    final = [w for w in collected if not any(o != w and o.endswith(w) for o in cset)]
    blocks = []
    for w in final:
        occ = [occurrences(s, w) for s in seqs]
        if all(len(o) == 1 for o in occ):
            blocks.append({"w": w, "depth": len(w), "pos": [o[0] for o in occ]})
    return blocks


def dfs_key(seqs, w):
    """Order in which collectNodes reaches the node of w: children of a node are visited in the
    order their first characters were first inserted under that node."""
    key = []
    for length in range(1, len(w) + 1):
        key.append(first_end(seqs, w[:length]))
    return key


def order_blocks(seqs, blocks):
    """insertSortedItem: decreasing depth; among equal depths the later inserted comes first."""
    seq = sorted(blocks, key=lambda b: dfs_key(seqs, b["w"]))        # insertion (DFS) order
    out = []
    for b in seq:
        i = 0
        while i < len(out) and b["depth"] < out[i]["depth"]:
            i += 1
        out.insert(i, b)
    return out


def link_chains(seqs, blocks):
    for b in blocks:
        b["size"] = 0
        b["total"] = 0
        b["next"] = None
    for k, s in enumerate(seqs):
        n = len(s)
        limit = n
        prev = None
        for b in sorted(blocks, key=lambda x: x["pos"][k]):
            p = b["pos"][k]
            if p + b["depth"] >= limit:
                continue
            if prev is not None:
                if prev["size"] == 0:
                    if prev["next"] is None:
                        prev["next"] = b
                    elif prev["next"] is not b:
                        prev["next"] = None
                        prev["size"] = -1
            else:
                limit += p
            prev = b
    for b in blocks:                                           # csamsa.c:181-224
        if b["total"] == -1:
            continue
        b["size"] = b["depth"]
        prev = b
        cur = b["next"]
        guard = 0
        while cur is not None:
            guard += 1
            if guard > 4 * len(blocks) + 8:
                raise RuntimeError("chain accumulation does not terminate (the reference loops here too)")
            interval = min(((cur["pos"][k] - (prev["pos"][k] + prev["depth"])) + (len(seqs[k]) if cur["pos"][k] < prev["pos"][k] else 0))
                           for k in range(len(seqs)))
            if cur["total"] > 0:
                b["size"] += cur["size"]
                b["total"] += cur["total"]
                b["total"] += interval
                cur["size"] = cur["depth"]
                cur["total"] = -1
                break
            cur["size"] = cur["depth"]
            b["size"] += cur["size"]
            b["total"] += interval
            cur["total"] = -1
            prev = cur
            cur = cur["next"]
        b["total"] += b["size"]
    return blocks


def sort_list(blocks):
    """nodeslinkedlists.c:55-79: repeated selection of the first strictly largest size."""
    lst = list(blocks)
    i = 0
    while i < len(lst):
        m = i
        for j in range(i + 1, len(lst)):
            if lst[j]["size"] > lst[m]["size"]:
                m = j
        if m != i:
            lst.insert(i, lst.pop(m))
        else:
            i += 1
    return lst


def analyze(seqs):
    blocks = order_blocks(seqs, find_blocks(seqs))
    link_chains(seqs, blocks)
    return sort_list(blocks)


def rotations(seqs):
    lst = analyze(seqs)
    return list(lst[0]["pos"]) if lst else None
