"""Rotation finder, CPU tests: csadp_find_rotations against the rotation
offsets the reference printed for its own example sets, against the brute-force oracle
(oracle/rot_oracle.py) and -- where oracle/_ref is present -- against the reference's tree
analysis itself, block list included."""
import os
import sys

import pytest

import csa_amd
from helpers import GOLDEN, ROOT, have_ref, load_golden, read_fasta, ref_rotations, rng, rotated_family

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rot_oracle  # noqa: E402

ROT = {"Primates": [1947, 1949, 1950, 2530, 1952, 1946, 1951, 1952, 1975, 1955, 1954, 2475, 1948, 1947, 1940, 1948],
       "Mammals": [1283, 1304, 1263, 1640, 1277, 1722, 1295, 1272, 1851, 1273, 1266, 1273],
       "Set3": [2405, 2407, 2408, 2988, 2412, 2404, 2409, 2405, 2451, 2412, 2420, 2936, 2408, 2402, 2400, 2406, 2392, 3709, 5471]}


@pytest.mark.parametrize("name", ["Primates", "Mammals", "Set3"])
def test_example_sets_reproduce_reference_rotations(name):
    """offsets in the headers of the reference's <set>-Rotated.fasta (mode R)."""
    _, seqs = read_fasta(os.path.join(GOLDEN, "data", name + ".txt"))
    rc, rot, info = csa_amd.find_rotations(seqs)
    assert rc == 0 and rot == ROT[name]
    assert rot == load_golden("pipeline.json")[name]["rotations"]       # parsed from the reference's own -Rotated.fasta
    if name == "Set3":       # 19 sequences incl. two distant ones: five common blocks only, which is why its anchoring fails
        assert info["blocks"] == 5 and info["chain_size"] == 23
    else:
        assert info["blocks"] > 40 and info["chain_size"] > 100
    # every rotated sequence starts with the same block
    d = info["first_block_depth"]
    heads = {(s[r:] + s[:r])[:d] for s, r in zip(seqs, rot)}
    assert len(heads) == 1


def _small_cases(seed, count):
    r = rng(seed)
    out = []
    while len(out) < count:
        nseq = r.choice([2, 2, 3, 4])
        seqs = rotated_family(r, nseq, r.choice([30, 50, 80]), mut=r.choice([0.02, 0.05, 0.1]), indel=r.choice([0.0, 0.03]))
        if any(len(a) == len(b) and b in a + a for i, a in enumerate(seqs) for b in seqs[i + 1:]):
            continue            # identical rotations: the reference discards the sequence
        out.append(seqs)
    return out


def test_product_matches_bruteforce_oracle():
    checked = loops = 0
    for seqs in _small_cases(11, 250):
        minlen = min(len(s) for s in seqs)
        try:
            lst = rot_oracle.analyze(seqs)
        except RuntimeError:
            rc, _, _ = csa_amd.find_rotations(seqs)     # the reference's bookkeeping does not terminate
            assert rc == csa_amd.ERR_RANGE
            loops += 1
            continue
        if any(b["depth"] >= minlen for b in lst):
            continue            # leaf-depth blocks are outside the product's definition
        rc, rot, info = csa_amd.find_rotations(seqs)
        if not lst:
            assert rc == csa_amd.ERR_RANGE
            continue
        assert rc == 0, seqs
        assert rot == lst[0]["pos"], seqs
        assert info["blocks"] == len(lst) and info["chain_size"] == lst[0]["size"] and info["chain_span"] == lst[0]["total"]
        checked += 1
    assert checked > 150


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libcsa_ref.so not built (needs /root/reference)")
def test_oracle_matches_reference_tree_analysis():
    """Pins oracle/rot_oracle.py: same blocks in the same list order with the same chain sizes
    as the reference's blockslist after analyzeTree (csamsa.c:271-308)."""
    checked = 0
    for seqs in _small_cases(5, 160):
        minlen = min(len(s) for s in seqs)
        try:
            lst = rot_oracle.analyze(seqs)
        except RuntimeError:
            assert ref_rotations(seqs, timeout=3)[0] == -9        # the reference really loops
            continue
        if not lst or any(b["depth"] >= minlen for b in lst):
            continue
        rc, rot, blocks = ref_rotations(seqs)
        assert rc == 0
        assert rot == lst[0]["pos"]
        assert blocks == [(b["depth"], b["size"], b["total"], b["pos"]) for b in lst]
        checked += 1
    assert checked > 100


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libcsa_ref.so not built (needs /root/reference)")
def test_product_matches_reference_on_medium_families():
    r = rng(2026)
    checked = 0
    for nseq, length in [(3, 600), (5, 1500), (8, 2500), (12, 1200), (4, 4000), (16, 800), (6, 3000), (2, 5000)]:
        for _ in range(3):
            seqs = rotated_family(r, nseq, length, mut=r.choice([0.03, 0.08, 0.15]), indel=r.choice([0.005, 0.02]))
            ref_rc, ref_rot, blocks = ref_rotations(seqs)
            rc, rot, info = csa_amd.find_rotations(seqs)
            if ref_rc == -9:
                assert rc == csa_amd.ERR_RANGE
                continue
            if ref_rc == -8:
                assert rc != 0
                continue
            assert ref_rc == 0 and rc == 0
            assert rot == ref_rot, (nseq, length)
            assert info["blocks"] == len(blocks) and info["chain_size"] == blocks[0][1]
            checked += 1
    assert checked >= 15


def test_rotation_finder_argument_errors():
    assert csa_amd.find_rotations([b"ACGT"])[0] == csa_amd.ERR_ARG
    assert csa_amd.find_rotations([b"A", b"ACGT"])[0] == csa_amd.ERR_ARG
    # nothing in common
    assert csa_amd.find_rotations([b"AAAAAAAAAA", b"CCCCCCCCCC"])[0] == csa_amd.ERR_RANGE
