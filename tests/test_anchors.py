"""Anchor stage: oracle/anchor_oracle.py pinned against the compiled reference
(live when oracle/_ref is present, and through tests/golden/anchors.json everywhere), and the
product's csadp_build_anchor_map checked against both.  Host code only: runs without a GPU."""
import json
import os
import sys

import pytest

import helpers as H

sys.path.insert(0, os.path.join(H.ROOT, "oracle"))
import anchor_oracle as A  # noqa: E402

import csa_amd  # noqa: E402


def golden():
    with open(os.path.join(H.ROOT, "tests", "golden", "anchors.json")) as f:
        return json.load(f)


def as_tuples(segs):
    return [(s[0], s[1], list(s[2])) for s in segs]


def fuzz_case(seed):
    r = H.rng(1000 + seed)
    n = r.choice([2, 3, 4, 6, 8])
    length = r.choice([40, 100, 200, 350])
    alpha = r.choice([b"ACGT", b"ACGT", b"AC", b"ACG", b"ACGTNRY"])
    fam = H.random_family(r, n, length, mut=r.choice([0.02, 0.08, 0.2]), indel=r.choice([0.0, 0.03, 0.08]), alphabet=alpha)
    if r.random() < 0.4:
        fam = [f + f[:len(f) // 3] for f in fam]                      # tandem repeats: several occurrences per node
    fam = [f if len(f) >= 12 else f + b"ACGTTGCAAGCT" for f in fam]
    rots = [r.randrange(len(f)) if r.random() < 0.7 else 0 for f in fam]
    return fam, rots, alpha


def test_oracle_matches_golden_maps():
    g = golden()
    assert len(g["families"]) >= 60
    for case in g["families"]:
        seqs = [s.encode() for s in case["seqs"]]
        border, segs = A.alignment_map(seqs, case["rotations"])
        assert len(border) == case["border_nodes"]
        assert segs == as_tuples(case["segments"])


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built")
def test_oracle_matches_reference_live():
    checked = 0
    for seed in range(60):
        fam, rots, alpha = fuzz_case(seed)
        if alpha == b"ACGTNRY":
            continue                       # the reference's DP misbehaves on non-ACGT gaps (survey quirk Q4)
        try:
            border, segs = A.alignment_map(fam, rots)
        except ValueError:
            continue
        rc, rot, rborder, rsegs = H.ref_alignment_map(fam, given_rot=rots, timeout=30)
        if rc != 0:
            continue
        assert sorted(rborder, key=lambda b: b[1][0][0]) == border, seed
        assert rsegs == segs, seed
        checked += 1
    assert checked >= 30


def test_product_matches_golden_maps():
    g = golden()
    for case in g["families"]:
        rc, segs, nodes = csa_amd.build_anchor_map([s.encode() for s in case["seqs"]], case["rotations"])
        assert rc == 0
        assert nodes == case["border_nodes"]
        assert segs == as_tuples(case["segments"])


@pytest.mark.parametrize("name", ["Primates", "Mammals", "Set3"])
def test_product_matches_golden_example_sets(name):
    want = golden()["sets"][name]
    _, seqs = H.read_fasta(os.path.join(H.ROOT, "tests", "golden", "data", name + ".txt"))
    rc, segs, nodes = csa_amd.build_anchor_map(seqs, want["rotations"])
    assert rc == 0
    assert nodes == want["border_nodes"]
    assert segs == as_tuples(want["segments"])
    assert sum(s[1] for s in segs) == H.load_golden("pipeline.json")[name]["dp_calls"]


def test_product_matches_oracle_fuzz():
    checked = 0
    for seed in range(200):
        fam, rots, _ = fuzz_case(seed)
        rc, segs, nodes = csa_amd.build_anchor_map(fam, rots)
        try:
            border, want = A.alignment_map(fam, rots)
        except ValueError:
            assert rc == csa_amd.ERR_RANGE
            continue
        assert rc == 0, seed
        assert nodes == len(border), seed
        assert segs == want, seed
        checked += 1
    assert checked >= 190


def test_suffix_equal_to_a_whole_rotation_is_rejected():
    short = b"GAAGCGAAAGAAGCGTGCGACTATGAACGGGGCCCTGTG"
    cut = 17
    longer = b"C" + short[cut:] + short[:cut]                     # one inserted letter in front of a rotation
    other = b"AAGAACGTGCGACTATGACGGGGCCCTGTGCGAAGCGA"
    with pytest.raises(ValueError):
        A.alignment_map([other, short, longer], [0, 0, 0])
    rc, _, _ = csa_amd.build_anchor_map([other, short, longer], [0, 0, 0])
    assert rc == csa_amd.ERR_RANGE


def test_argument_errors():
    assert csa_amd.build_anchor_map([b"ACGT"], [0])[0] == csa_amd.ERR_ARG
    assert csa_amd.build_anchor_map([b"ACGTACGT", b"ACGTTCGT"], [0, 8])[0] == csa_amd.ERR_ARG
    assert csa_amd.build_anchor_map([b"ACGTACGT", b"ACGTTCGT"], [-1, 0])[0] == csa_amd.ERR_ARG
