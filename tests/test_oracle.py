"""CPU tests of the CHECKER: the oracle (oracle/csa_dp_oracle.c) against the golden vectors
generated from the compiled reference, against the reference library itself when it is
present (oracle/_ref, build container / prebuilt on the GPU box), and against the values
quoted in SURVEY.md 8(c)."""
import ctypes
import os

import numpy as np
import pytest

from helpers import (GOLDEN, fnv1a, golden_aligned, golden_task, have_ref, load_golden, oracle_lib,
                     oracle_progressive, random_family, read_fasta, ref_progressive, rng, sp_score)


@pytest.mark.parametrize("name", ["tiny_pairs.json", "tiny_families.json"])
def test_oracle_matches_golden_strings(name):
    cases = load_golden(name)
    assert len(cases) >= 160
    for c in cases:
        texts, rots, starts, ends = golden_task(c)
        cons, strs, _ = oracle_progressive(texts, rots, starts, ends)
        assert cons == c["consensus"]
        assert strs == golden_aligned(c)


def test_oracle_real_pair_survey_values():
    """Primates seq0 x seq1, rotations 1947/1949: len 16589, SP 15197, FNV-1a 7ee50a99."""
    _, seqs = read_fasta(os.path.join(GOLDEN, "data", "Primates.txt"))
    cons, strs, st = oracle_progressive([seqs[0], seqs[1]], [1947, 1949])
    assert cons == 16589 and st.last_score == 15197 and sp_score(strs) == 15197
    assert "%08x" % fnv1a(strs) == "7ee50a99"
    assert st.cells == 16554 * 16563 and st.fills == 1 and st.stale_border_fills == 0
    gold = [c for c in load_golden("real_pairs.json") if c["set"] == "Primates" and (c["a"], c["b"]) == (0, 1)
            and c["rots"] == [1947, 1949]][0]
    assert gold["consensus"] == cons and gold["sp"] == 15197 and gold["fnv1a"] == "7ee50a99"


def test_golden_real_pairs_complete():
    real = load_golden("real_pairs.json")
    assert sum(1 for c in real if c["set"] == "Primates") == 121      # 120 pairs + (0,1) unrotated
    assert sum(1 for c in real if c["set"] == "Mammals") == 66
    cells = sum(c["len_a"] * c["len_b"] for c in real if c["set"] == "Mammals")
    assert cells == 18593884141


def test_oracle_alphabet_and_argument_errors():
    assert oracle_progressive([b"ACGT", b"ACNT"])[0] == -2
    assert oracle_progressive([b"ACGT", b"ACGT"], [0, 0], [0, 0], [5, 4])[0] == -1
    cons, strs, _ = oracle_progressive([b"ACGT", b"ACGT"], [0, 0], [2, 1], [2, 1])
    assert cons == 0 and strs == [None, None]


def test_oracle_stale_border_quirk_is_exercised():
    """Equal-length regions trigger the un-refreshed borders of dynamicprogramming.c:957."""
    r = rng(3)
    stale = 0
    for _ in range(30):
        fam = random_family(r, 5, 40, mut=0.2, indel=0.0)
        _, _, st = oracle_progressive(fam)
        stale += st.stale_border_fills
    assert stale > 0


def test_odp_fill_direction_priority():
    """diag >= up && diag >= left -> D; else left >= up -> L; else U (:1014-1025)."""
    lib = oracle_lib()
    nrows, ncols = 3, 3
    sv = (ctypes.c_int * ((ncols + 1) * 5))()
    for k, ch in enumerate(b"ACG", start=1):
        sv[k * 5 + b"ACGT".index(ch)] = 1
    rows = (ctypes.c_byte * nrows)(0, 1, 2)
    H = (ctypes.c_int * 16)()
    D = ctypes.create_string_buffer(16)
    assert lib.odp_fill(nrows, ncols, rows, sv, 1, None, 1, H, D) == 0
    Hm = np.array(list(H)).reshape(4, 4)
    assert Hm[3, 3] == 3 and D.raw[5] == ord("D") and D.raw[10] == ord("D") and D.raw[15] == ord("D")
    assert list(Hm[0]) == [0, -1, -2, -3] and list(Hm[:, 0]) == [0, -1, -2, -3]
    assert D.raw[6] == ord("L") or D.raw[6] == ord("D")   # (1,2): diag -1-1=-2, left 1-1=0 -> L
    assert D.raw[6] == ord("L")
    assert D.raw[9] == ord("U")                            # (2,1): up 1-1=0 beats diag -2 and left -3


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libcsa_ref.so not built (needs /root/reference)")
def test_oracle_vs_compiled_reference_fuzz():
    r = rng(11)
    for it in range(400):
        n = r.choice([2, 2, 3, 4, 5, 6, 8])
        length = r.choice([0, 1, 2, 3, 5, 8, 13, 21, 40, 64])
        fam = random_family(r, n, length, mut=r.choice([0.0, 0.1, 0.3, 0.9]), indel=r.choice([0.0, 0.1, 0.3]))
        fam = [f if f else b"A" for f in fam]
        rots = [r.randrange(len(f)) for f in fam]
        starts, ends = [], []
        for f in fam:
            a = r.randrange(len(f) + 1)
            b = r.randrange(a, len(f) + 1)
            if r.random() < 0.6:
                a, b = 0, len(f)
            starts.append(a)
            ends.append(b)
        c1, s1, _ = oracle_progressive(fam, rots, starts, ends)
        c2, s2, _ = ref_progressive(fam, rots, starts, ends)
        assert c1 == c2 and s1 == s2, (it, fam, rots, starts, ends)


def test_oracle_reproduces_benchmark_pair_digests():
    """Two pairs of the benchmark workload (config 4): the oracle's strings carry the digests the
    compiled reference produced (tests/golden/config4_pairs.json)."""
    from helpers import oracle_progressive, sp_score, synth_pair
    for g in load_golden("config4_pairs.json")[:2]:
        a, b, ra, rb = synth_pair(g["pair"])
        cons, strs, st = oracle_progressive([a, b], [ra, rb])
        assert (cons, sp_score(strs), "%08x" % fnv1a(strs)) == (g["consensus"], g["sp"], g["fnv1a"])
        assert st.last_score == g["sp"]


def test_oracle_reproduces_a_sample_of_the_wide_reference_fixtures():
    """Round 5's reference-held records -- all 1024 pairs of config 4, the 64 unrelated 16 kbp pairs, the 171 pairs of config 5 the
    reference's matrices can hold (tests/golden/make_golden.py --config4-all / --unrelated / --config5) -- pin the ORACLE too: a sample
    of each, string digests included; the linear-space score on a wider sample."""
    from helpers import oracle_pair_score_linear, oracle_progressive, sp_score, synth_pair
    from csa_amd.synth import config5_lengths
    g4 = load_golden("config4_all.json")
    assert g4["pairs"] == 1024 and len(g4["fnv1a"]) == len(g4["sp"]) == len(g4["consensus"]) == 1024
    for g in load_golden("config4_pairs.json"):                    # the round-2 fixture is a prefix of the new one
        assert (g4["consensus"][g["pair"]], g4["sp"][g["pair"]], g4["fnv1a"][g["pair"]]) == (g["consensus"], g["sp"], g["fnv1a"])
    for p in (517, 1023):
        a, b, ra, rb = synth_pair(p)
        cons, strs, st = oracle_progressive([a, b], [ra, rb])
        assert (cons, sp_score(strs), "%08x" % fnv1a(strs)) == (g4["consensus"][p], g4["sp"][p], g4["fnv1a"][p])
    for p in (100, 333, 640, 900):
        a, b, ra, rb = synth_pair(p)
        assert oracle_pair_score_linear([a, b], [ra, rb]) == g4["sp"][p]
    gu = load_golden("unrelated_pairs.json")
    assert len(gu) == 64
    for g in (gu[5], gu[63]):
        a, b, ra, rb = synth_pair(70000 + g["pair"], unrelated=True)
        cons, strs, st = oracle_progressive([a, b], [ra, rb])
        assert (cons, sp_score(strs), "%08x" % fnv1a(strs)) == (g["consensus"], g["sp"], g["fnv1a"])
    g5 = load_golden("config5_pairs.json")
    la, _ = config5_lengths(256)
    assert len(g5) == sum(1 for x in la if x <= 40000) >= 170
    small = sorted(g5, key=lambda g: g["len_a"])
    for g in small[:6] + small[len(small) // 2:len(small) // 2 + 3]:           # 1 k .. 7 k letters: full strings
        a, b, ra, rb = synth_pair(20000 + g["index"], length=int(la[g["index"]]))
        assert (len(a), len(b)) == (g["len_a"], g["len_b"])
        cons, strs, st = oracle_progressive([a, b], [ra, rb])
        assert (cons, sp_score(strs), "%08x" % fnv1a(strs)) == (g["consensus"], g["sp"], g["fnv1a"])
    for g in small[-2:]:                                                        # the two longest (38 k letters): the score
        a, b, ra, rb = synth_pair(20000 + g["index"], length=int(la[g["index"]]))
        assert oracle_pair_score_linear([a, b], [ra, rb]) == g["sp"]


def test_linear_space_score_equals_the_full_matrix():
    """odp_pair_score_linear (two rows, no directions) == dpmatrix[nrows][ncols] of the full
    restatement on fuzzed pairs (sub-regions, rotations, empty sides) ..."""
    from helpers import oracle_pair_score_linear
    r = rng(31337)
    for _ in range(200):
        fam = random_family(r, 2, r.choice([1, 7, 33, 64, 200, 700]), mut=r.choice([0.0, 0.1, 0.6]), indel=r.choice([0.0, 0.1, 0.3]))
        fam = [f if f else b"C" for f in fam]
        rots = [r.randrange(len(f)) for f in fam]
        starts = [r.randrange(len(f) + 1) if r.random() < 0.3 else 0 for f in fam]
        ends = [r.randrange(a, len(f) + 1) if r.random() < 0.3 else len(f) for a, f in zip(starts, fam)]
        rc, strs, st = oracle_progressive(fam, rots, starts, ends)
        if strs[0] is None:
            continue                      # both regions empty: the reference returns before any fill (:916)
        assert oracle_pair_score_linear(fam, rots, starts, ends) == st.last_score == sp_score(strs)


def test_linear_space_score_equals_reference_goldens():
    """... and == the SP score of the COMPILED REFERENCE's strings on whole mitochondrial genomes
    (real_pairs.json) and on the benchmark workload (config4_pairs.json)."""
    from helpers import oracle_pair_score_linear, read_fasta, GOLDEN
    from csa_amd.synth import synth_pair
    import os
    for g in load_golden("config4_pairs.json")[:6]:
        a, b, ra, rb = synth_pair(g["pair"])
        assert oracle_pair_score_linear([a, b], [ra, rb]) == g["sp"]
    sets = {}
    for g in load_golden("real_pairs.json")[::23]:
        if g["set"] not in sets:
            sets[g["set"]] = read_fasta(os.path.join(GOLDEN, "data", g["set"] + ".txt"))[1]
        seqs = sets[g["set"]]
        assert oracle_pair_score_linear([seqs[g["a"]], seqs[g["b"]]], g["rots"]) == g["sp"]


def test_oracle_sp_stats_equal_reference_mode_s():
    """odp_sp_stats restates tools.c:194-293; sp_stats.json holds what the reference PROGRAM printed
    in mode S for the same rows (100 alignments incl. all-gap columns and IUPAC letters)."""
    from helpers import oracle_sp_stats
    for case in load_golden("sp_stats.json"):
        assert oracle_sp_stats(case["rows"]) == case["mode_s"]
