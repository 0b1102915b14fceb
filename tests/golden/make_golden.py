#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the COMPILED REFERENCE
(oracle/_ref/libcsa_ref.so, built by `make -C oracle _ref` from the unmodified
sources under /root/reference/source).  Only runs in the build container.

Outputs (all data, no reference source text):
  data/Primates.txt, data/Mammals.txt   the reference's own example inputs
                                        (Manual/*.txt) with CRLF normalised
  tiny_pairs.json      >=200 random + adversarial 2-sequence cases, full strings
  tiny_families.json   N=3..8 progressive cases (DeleteGappedColumns, Q1), full strings
  real_pairs.json      whole-sequence pairs of Primates/Mammals: length, SP score,
                       FNV-1a digest (the values quoted in SURVEY.md 8c included)
  pipeline.json        (--pipeline) md5 of <set>-Aligned.fasta / -Rotated.fasta written by the
                       unmodified reference program in mode N (oracle/_ref/CSA_ref), the rotations of its
                       -Rotated.fasta headers, and the four statistics its mode S (tools.c:194-293,
                       CalculateSumOfPairsScore) prints for that -Aligned.fasta.  Sets: Primates, Mammals
                       and Set3 (website/Examples.zip; the one input with LARGE profile fills)
  sp_stats.json        (--sp) mode S of the reference program on small alignments (the rows of
                       anchors.json's families + adversarial column patterns): rows and the four numbers

  config4_pairs.json   (--config4) the first 32 pairs of the benchmark workload (SURVEY 8d config 4,
                       csa_amd/synth.py) through the compiled reference: length, SP score, FNV-1a
  config4_all.json     (--config4-all) ALL 1024 pairs of config 4 through the compiled reference (8 processes):
                       parallel arrays consensus / sp / fnv1a indexed by pair number
  config5_pairs.json   (--config5) every pair of config 5 (csa_amd/synth.py:config5_lengths, pair i =
                       synth_pair(20000 + i, length=la[i])) whose longer side is <= 40 000 letters -- what the
                       reference's 5 B/cell matrices allow here -- index, lengths, consensus, sp, fnv1a
  unrelated_pairs.json (--unrelated) the 64 UNRELATED 16 kbp pairs of config 4's second variant
                       (synth_pair(70000 + p, unrelated=True)): consensus, sp, fnv1a
  anchors.json         (--anchors) the anchor stage of the compiled reference (ref_shim.c:
                       csa_ref_alignment_map): small families with their rotations, border-node
                       count, final alignment map and the rows SaveAlignment wrote; and the
                       alignment maps of the two example sets

usage: python tests/golden/make_golden.py [--all-pairs]
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import (fnv1a, random_family, read_fasta, ref_alignment_map, ref_progressive, rng,  # noqa: E402
                     rotated_family, sp_score)

REF_MANUAL = "/root/reference/Manual"
# rotations printed by the reference's mode R for each example set
ROT = {
    "Primates": [1947, 1949, 1950, 2530, 1952, 1946, 1951, 1952, 1975, 1955, 1954, 2475, 1948, 1947, 1940, 1948],
    "Mammals": [1283, 1304, 1263, 1640, 1277, 1722, 1295, 1272, 1851, 1273, 1266, 1273],
    "Set3": [2405, 2407, 2408, 2988, 2412, 2404, 2409, 2405, 2451, 2412, 2420, 2936, 2408, 2402, 2400, 2406, 2392, 3709, 5471],
}


def case(texts, rots, starts, ends):
    cons, strs, _ = ref_progressive(texts, rots, starts, ends)
    return {
        "texts": [t.decode() for t in texts], "rots": rots, "starts": starts, "ends": ends,
        "consensus": cons,
        "aligned": [s.decode() if s is not None else None for s in strs],
    }


def tiny_pairs():
    r = rng(20261003)
    cases = []
    adversarial = [
        (b"A", b"A"), (b"A", b"C"), (b"AAAAAAAA", b"AAAA"), (b"ACGTACGT", b"TGCATGCA"),
        (b"AAAAAAAAAAAAAAAA", b"CCCCCCCCCCCCCCCC"), (b"ACGT" * 8, b"ACGT" * 8),
        (b"ACGT" * 8, b"CGTA" * 8), (b"A" * 33, b"A" * 31 + b"C"), (b"GATTACA", b"GCATGCT"),
        (b"ACACACACAC", b"CACACACACA"), (b"T", b"ACGTACGTACGT"), (b"ACGTACGTACGT", b"G"),
    ]
    for a, b in adversarial:
        cases.append(case([a, b], [0, 0], [0, 0], [len(a), len(b)]))
        cases.append(case([b, a], [0, 0], [0, 0], [len(b), len(a)]))
    # empty regions (one side, both sides)
    cases.append(case([b"ACGT", b"ACGT"], [0, 0], [2, 0], [2, 4]))
    cases.append(case([b"ACGT", b"ACGT"], [0, 0], [0, 1], [4, 1]))
    cases.append(case([b"ACGT", b"ACGT"], [1, 2], [3, 3], [3, 3]))
    while len(cases) < 240:
        la = r.choice([1, 2, 3, 5, 8, 13, 21, 34, 48, 64])
        fam = random_family(r, 2, la, mut=r.choice([0.0, 0.05, 0.2, 0.75]), indel=r.choice([0.0, 0.1, 0.3]))
        fam = [f if f else b"G" for f in fam]
        rots = [r.randrange(len(f)) for f in fam]
        starts, ends = [], []
        for f in fam:
            if r.random() < 0.7:
                starts.append(0)
                ends.append(len(f))
            else:
                a = r.randrange(len(f) + 1)
                starts.append(a)
                ends.append(r.randrange(a, len(f) + 1))
        cases.append(case(fam, rots, starts, ends))
    return cases


def tiny_families():
    r = rng(77)
    cases = []
    while len(cases) < 160:
        n = r.choice([3, 3, 4, 5, 6, 8])
        base = r.choice([4, 9, 17, 30, 48])
        fam = random_family(r, n, base, mut=r.choice([0.0, 0.1, 0.3]), indel=r.choice([0.0, 0.15, 0.35]))
        fam = [f if f else b"T" for f in fam]
        if r.random() < 0.3:      # equal lengths trigger the stale-border rule (Q1)
            m = min(len(f) for f in fam)
            fam = [f[:m] for f in fam]
        rots = [r.randrange(len(f)) for f in fam]
        starts = [0] * n
        ends = [len(f) for f in fam]
        if r.random() < 0.2:      # one empty region
            k = r.randrange(n)
            starts[k] = ends[k] = r.randrange(len(fam[k]) + 1)
        cases.append(case(fam, rots, starts, ends))
    return cases


def real_pairs(all_pairs):
    out = []
    for name in ("Primates", "Mammals"):
        _, seqs = read_fasta(os.path.join(HERE, "data", name + ".txt"))
        n = len(seqs)
        if all_pairs:
            pairs = [(a, b) for a in range(n) for b in range(a + 1, n)]
        elif name == "Primates":
            pairs = [(0, 1), (2, 14), (9, 10), (3, 11)]
        else:
            pairs = [(0, 1), (5, 8)]
        for a, b in pairs:
            for rots in ([ROT[name][a], ROT[name][b]],) + (([0, 0],) if (name, a, b) == ("Primates", 0, 1) else ()):
                cons, strs, sec = ref_progressive([seqs[a], seqs[b]], list(rots))
                out.append({"set": name, "a": a, "b": b, "rots": list(rots), "consensus": cons,
                            "sp": sp_score(strs), "fnv1a": "%08x" % fnv1a(strs),
                            "len_a": len(seqs[a]), "len_b": len(seqs[b])})
                print(name, a, b, rots, cons, out[-1]["sp"], out[-1]["fnv1a"], "%.1fs" % sec, flush=True)
    return out


def config4_pairs(n=32):
    from csa_amd.synth import synth_pair
    out = []
    for p in range(n):
        a, b, ra, rb = synth_pair(p)
        cons, strs, sec = ref_progressive([a, b], [ra, rb])
        out.append({"pair": p, "consensus": cons, "sp": sp_score(strs), "fnv1a": "%08x" % fnv1a(strs)})
        print("config4 pair", p, cons, out[-1]["sp"], out[-1]["fnv1a"], "%.1fs" % sec, flush=True)
    return out


def _ref_digest(job):
    """(key, texts, rots) -> (key, consensus, sp, fnv1a, seconds); runs in a worker process (the reference is all globals)."""
    key, kind = job
    from csa_amd.synth import config5_lengths, synth_pair
    if kind == "config4":
        a, b, ra, rb = synth_pair(key)
    elif kind == "unrelated":
        a, b, ra, rb = synth_pair(70000 + key, unrelated=True)
    else:
        la, _ = config5_lengths(256)
        a, b, ra, rb = synth_pair(20000 + key, length=int(la[key]))
    cons, strs, sec = ref_progressive([a, b], [ra, rb])
    return key, len(a), len(b), cons, sp_score(strs), "%08x" % fnv1a(strs), sec


def _run_pool(jobs, workers, label):
    """The compiled reference keeps 5 bytes per cell (dynamicprogramming.c:964-981): `workers` bounds the memory in use."""
    import multiprocessing as mp
    out = {}
    with mp.get_context("fork").Pool(workers, maxtasksperchild=4) as pool:
        for key, la, lb, cons, sp, dig, sec in pool.imap_unordered(_ref_digest, jobs):
            out[key] = (la, lb, cons, sp, dig)
            print(label, key, la, lb, cons, sp, dig, "%.1fs" % sec, "(%d/%d)" % (len(out), len(jobs)), flush=True)
    return out


def config4_all(n=1024, workers=8):
    got = _run_pool([(p, "config4") for p in range(n)], workers, "config4")
    return {"pairs": n, "consensus": [got[p][2] for p in range(n)], "sp": [got[p][3] for p in range(n)],
            "fnv1a": [got[p][4] for p in range(n)]}


def unrelated_pairs(n=64, workers=8):
    got = _run_pool([(p, "unrelated") for p in range(n)], workers, "unrelated")
    return [{"pair": p, "consensus": got[p][2], "sp": got[p][3], "fnv1a": got[p][4]} for p in range(n)]


def config5_pairs(limit=40000):
    from csa_amd.synth import config5_lengths
    la, _ = config5_lengths(256)
    # the partner is within a few per cent of la: 1.06 la bounds it
    small = [i for i in range(256) if la[i] * 1.06 <= 20000]
    large = [i for i in range(256) if 20000 < la[i] * 1.06 and la[i] * 1.06 <= limit * 1.06]
    got = _run_pool([(i, "config5") for i in small], 8, "config5")
    got.update(_run_pool([(i, "config5") for i in sorted(large, key=lambda i: -la[i])], 3, "config5"))   # up to 8.5 GB each
    keep = sorted(i for i in got if max(got[i][0], got[i][1]) <= limit)
    return [{"index": i, "len_a": got[i][0], "len_b": got[i][1], "consensus": got[i][2], "sp": got[i][3], "fnv1a": got[i][4]}
            for i in keep]


REF_BIN = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "CSA_ref")


def ref_mode_s(path):
    """The four numbers the reference program prints in mode S (csamsa.c:639-641 -> tools.c:194-293)."""
    import re
    import subprocess
    with open(os.devnull) as devnull:
        log = subprocess.run([REF_BIN, "S", os.path.basename(path)], cwd=os.path.dirname(path), stdin=devnull,
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode(errors="replace")
    m = re.search(r"Consensus size = (-?\d+)\s+Average gaps per sequence = (-?\d+)\s+"
                  r"Number of conserved columns = (-?\d+)\s+Sum-of-Pairs score = (-?\d+)", log)
    assert m, log
    return {"consensus": int(m.group(1)), "avg_gaps": int(m.group(2)), "conserved": int(m.group(3)), "sp": int(m.group(4))}


def pipeline():
    """Whole-program goldens: run the UNMODIFIED reference binary (oracle/_ref/CSA_ref, built by
    `make -C oracle _dropin`) in mode N on the example sets, record the md5 of its outputs, the
    rotations it chose and what its own mode S says about the alignment it wrote."""
    import hashlib
    import re
    import shutil
    import subprocess
    import tempfile
    out = {}
    for name in ("Primates", "Mammals", "Set3"):
        with tempfile.TemporaryDirectory() as tmp:
            shutil.copy(os.path.join(HERE, "data", name + ".txt"), tmp)
            with open(os.devnull) as devnull:
                log = subprocess.run([REF_BIN, name + ".txt"], cwd=tmp, stdin=devnull, stdout=subprocess.PIPE,
                                     stderr=subprocess.STDOUT).stdout.decode(errors="replace")
            rec = {}
            for kind in ("Aligned", "Rotated"):
                with open(os.path.join(tmp, "%s-%s.fasta" % (name, kind)), "rb") as f:
                    rec[kind.lower() + "_md5"] = hashlib.md5(f.read()).hexdigest()
            with open(os.path.join(tmp, name + "-Rotated.fasta"), "rb") as f:
                rec["rotations"] = [int(ln.rsplit(b"@", 1)[1]) for ln in f if ln.startswith(b">")]
            rec["dp_calls"] = log.count("[(")
            # the stdout tokens of every ProgressiveDP call (dynamicprogramming.c:917, :1156, :689, :1159)
            rec["dp_log"] = re.findall(r"\[\([^\]]*\]", log)
            rec["mode_s"] = ref_mode_s(os.path.join(tmp, name + "-Aligned.fasta"))
            out[name] = rec
            print(name, rec, flush=True)
    return out


def sp_stats():
    """Mode S of the reference program on small alignments: rows + the four numbers."""
    import tempfile
    with open(os.path.join(HERE, "anchors.json")) as f:
        fams = json.load(f)["families"]
    cases = [f["rows"] for f in fams if len(f["rows"]) >= 2 and len(f["rows"][0]) > 0]
    r = rng(4242)
    for _ in range(40):                     # adversarial columns: all-gap, all-equal, one odd letter, IUPAC letters
        n = r.choice([2, 3, 5, 9, 17, 33, 64])
        length = r.choice([1, 2, 7, 64, 65, 257, 1000])
        alphabet = r.choice(["ACGT-", "A-", "ACGTN-", "AC"])
        cases.append(["".join(r.choice(alphabet) for _ in range(length)) for _ in range(n)])
    out = []
    for rows in cases:
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "x.fasta")
            with open(path, "w") as f:
                for i, row in enumerate(rows):
                    f.write(">s%d\n%s\n" % (i, row))
            out.append({"rows": rows, "mode_s": ref_mode_s(path)})
    return out


def anchors():
    import tempfile
    out = {"families": [], "sets": {}}
    seed = 0
    while len(out["families"]) < 60:
        seed += 1
        r = rng(7000 + seed)
        n = r.choice([2, 3, 4, 5, 8])
        length = r.choice([60, 150, 300])
        mut, indel = r.choice([0.03, 0.08, 0.15]), r.choice([0.0, 0.02, 0.05])
        found = seed % 2 == 0                      # rotations from the reference's own finder, or given
        if found:
            fam = rotated_family(r, n, length, mut=mut, indel=indel)
            given = None
        else:
            fam = random_family(r, n, length, mut=mut, indel=indel, alphabet=r.choice([b"ACGT", b"ACG"]))
            if r.random() < 0.4:
                fam = [f + f[:len(f) // 3] for f in fam]
            fam = [f if len(f) >= 12 else f + b"ACGTTGCAAGCT" for f in fam]
            given = [r.randrange(len(f)) if r.random() < 0.7 else 0 for f in fam]
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "a.fasta")
            rc, rot, border, segs = ref_alignment_map(fam, given_rot=given, savepath=path, timeout=30)
            if rc != 0:
                continue                            # the reference exits / does not terminate on this input
            rows = [ln.rstrip(b"\n").decode() for ln in open(path, "rb") if not ln.startswith(b">")]
        out["families"].append({"seqs": [f.decode() for f in fam], "given": given is not None, "rotations": rot,
                                "border_nodes": len(border), "segments": segs, "rows": rows})
    out["sets"] = anchor_sets()
    return out


def anchor_sets():
    """Alignment maps of the example sets (Set3: only two gaps survive the anchoring, 36 fills)."""
    sets = {}
    for name in ("Primates", "Mammals", "Set3"):
        _, seqs = read_fasta(os.path.join(HERE, "data", name + ".txt"))
        rc, rot, border, segs = ref_alignment_map(seqs, timeout=900)
        assert rc == 0 and rot == ROT[name], (name, rc, rot)
        sets[name] = {"rotations": rot, "border_nodes": len(border), "segments": segs}
    return sets


def copy_data():
    import zipfile
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    for name in ("Primates", "Mammals"):
        with open(os.path.join(REF_MANUAL, name + ".txt"), "rb") as f:
            raw = f.read().replace(b"\r\n", b"\n")
        with open(os.path.join(HERE, "data", name + ".txt"), "wb") as f:
            f.write(raw)
    # Set3 (19 sequences) ships only inside the web front end's example archive
    with zipfile.ZipFile("/root/reference/website/Examples.zip") as z:
        raw = z.read("Set3.txt").replace(b"\r\n", b"\n")
    with open(os.path.join(HERE, "data", "Set3.txt"), "wb") as f:
        f.write(raw)


def main():
    all_pairs = "--all-pairs" in sys.argv
    copy_data()
    if "--pipeline" in sys.argv:
        with open(os.path.join(HERE, "pipeline.json"), "w") as f:
            json.dump(pipeline(), f, indent=1)
        return
    if "--sp" in sys.argv:
        with open(os.path.join(HERE, "sp_stats.json"), "w") as f:
            json.dump(sp_stats(), f, indent=0)
        return
    for flag, name, fn in (("--config4-all", "config4_all.json", config4_all), ("--config5", "config5_pairs.json", config5_pairs),
                           ("--unrelated", "unrelated_pairs.json", unrelated_pairs)):
        if flag in sys.argv:
            with open(os.path.join(HERE, name), "w") as f:
                json.dump(fn(), f, indent=0)
            return
    if "--config4" in sys.argv:
        with open(os.path.join(HERE, "config4_pairs.json"), "w") as f:
            json.dump(config4_pairs(), f, indent=1)
        return
    if "--anchor-sets" in sys.argv:          # keep the families, regenerate the example sets only
        with open(os.path.join(HERE, "anchors.json")) as f:
            cur = json.load(f)
        cur["sets"] = anchor_sets()
        with open(os.path.join(HERE, "anchors.json"), "w") as f:
            json.dump(cur, f, indent=0)
        return
    if "--anchors" in sys.argv:
        with open(os.path.join(HERE, "anchors.json"), "w") as f:
            json.dump(anchors(), f, indent=0)
        return
    if "--only-real" not in sys.argv:
        with open(os.path.join(HERE, "tiny_pairs.json"), "w") as f:
            json.dump(tiny_pairs(), f, indent=0)
        with open(os.path.join(HERE, "tiny_families.json"), "w") as f:
            json.dump(tiny_families(), f, indent=0)
    with open(os.path.join(HERE, "real_pairs.json"), "w") as f:
        json.dump(real_pairs(all_pairs), f, indent=1)


if __name__ == "__main__":
    main()
