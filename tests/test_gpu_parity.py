"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI (libcsadp.so), against
the committed golden vectors (generated from the compiled reference) and against the oracle
on seeded inputs; at full size through size-independent properties.

Bar: bit-exact -- integer scores, aligned strings byte for byte."""
import pytest

import csa_amd
from helpers import (degap, fnv1a, golden_aligned, golden_task, load_golden, oracle_pair_score_linear, oracle_progressive,
                     random_family, read_fasta, rng, rotated, sp_score, synth_pair, GOLDEN)
import os

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def device():
    csa_amd.init(device=0)
    yield


# First fills (every pairwise task, and step 1 of every progressive task) run the bit-parallel
# kernels by default (csadp_bits.hip); CSADP_BITS=0 sends them to the 32-bit cell-per-lane kernel that
# serves every other fill (csadp_cells.hip) -- the independent cross-check of the bit-parallel path.
# The environment is read at every batch layout, so the tests below that take `fill_mode` cover both
# families.  Batches of 2-sequence tasks in bit-parallel mode also pack their inputs and build their rows
# on the device (csadp_pairio.hip); "bits-hostio" switches that off: tables written and traces applied by
# the host, as for N sequences.  (The tiled 32-bit and packed-16 families of rounds 1-2 are gone.)
# The direction walk of the 32-bit kernel is one serial walk for matrices of fewer than 512 rows and band-parallel
# from there on (csadp_cells_tb.hip; CSADP_TB_BAND_MIN, read at every batch layout): "cells-banded" makes every
# matrix, a one-row one included, take the scout / resolve / emit / gather kernels, "cells" none.
@pytest.fixture(params=["bits", "bits-hostio", "cells", "cells-banded"])
def fill_mode(request, monkeypatch):
    if request.param.startswith("bits"):
        monkeypatch.setenv("CSADP_LONE_CELLS", "0")             # also for large pairs alone (FillBatch::layout would pick the cell-per-lane path)
    if request.param == "bits-hostio":
        monkeypatch.setenv("CSADP_DEVICE_IO", "0")
    elif request.param != "bits":
        monkeypatch.setenv("CSADP_BITS", "0")
        monkeypatch.setenv("CSADP_TB_BAND_MIN", "1" if request.param == "cells-banded" else "2000000000")
    return request.param


# Profile steps (i >= 2, stale borders) run the persistent cell-per-lane kernel (csadp_cells.hip).
@pytest.fixture(params=["cells", "cells-banded", "cells-default"])
def profile_mode(request, monkeypatch):
    if request.param != "cells-default":
        monkeypatch.setenv("CSADP_TB_BAND_MIN", "1" if request.param == "cells-banded" else "2000000000")
    return request.param


def _check_case(case, got):
    exp = golden_aligned(case)
    assert got["status"] == 0
    assert got["consensus"] == case["consensus"]
    if all(e is None for e in exp):
        assert got["aligned"] is None
    else:
        assert got["aligned"] == exp


def test_tiny_pairs_golden(fill_mode):
    cases = load_golden("tiny_pairs.json")
    got = csa_amd.align_batch([golden_task(c) for c in cases])
    for c, g in zip(cases, got):
        _check_case(c, g)


def test_tiny_families_golden(profile_mode):
    """N=3..8 progressive tasks: profile mode of the kernel, stale borders (Q1),
    DeleteGappedColumns on the host, zero-length regions."""
    cases = load_golden("tiny_families.json")
    got = csa_amd.align_batch([golden_task(c) for c in cases])
    for c, g in zip(cases, got):
        _check_case(c, g)


def test_tiny_one_by_one_matches_batch():
    cases = load_golden("tiny_pairs.json")[:40]
    for c in cases:
        _check_case(c, csa_amd.align_batch([golden_task(c)])[0])


def test_score_equals_sp_for_pairs(fill_mode):
    cases = [c for c in load_golden("tiny_pairs.json") if c["aligned"][0] is not None]
    got = csa_amd.align_batch([golden_task(c) for c in cases])
    for c, g in zip(cases, got):
        if all(len(t) > 0 for t in g["aligned"]) and c["ends"][0] > c["starts"][0] and c["ends"][1] > c["starts"][1]:
            assert g["score"] == sp_score(g["aligned"])


@pytest.mark.parametrize("shape", [(63, 1000), (1000, 63), (1023, 1025), (1024, 1024), (1025, 1023),
                                   (2049, 5000), (5000, 2049), (1, 3000), (3000, 1), (4097, 4095)])
def test_ragged_shapes_vs_oracle(shape):
    """Strip / tile boundary cases: lengths around multiples of 64*C and of the tile height."""
    r = rng(shape[0] * 100003 + shape[1])
    a = bytes(r.choice(b"ACGT") for _ in range(shape[0]))
    b = random_family(r, 1, shape[1], mut=0.2, indel=0.05)[0] if shape[1] > 4 else b"ACG"[:shape[1]]
    if shape[0] > 4 and shape[1] > 4:
        # make them related so the path wanders but stays near the diagonal band
        m = min(shape)
        b = (a[:m] if len(a) >= m else a) + bytes(r.choice(b"ACGT") for _ in range(max(0, shape[1] - m)))
        b = bytes(ch if r.random() > 0.1 else r.choice(b"ACGT") for ch in b)[:shape[1]]
    rots = [r.randrange(len(a)), r.randrange(len(b))]
    g = csa_amd.align_batch([([a, b], rots, None, None)])[0]
    cons, strs, st = oracle_progressive([a, b], rots)
    assert g["status"] == 0 and g["consensus"] == cons
    assert g["aligned"] == strs
    assert g["score"] == st.last_score


def test_unrelated_pair_negative_scores(fill_mode):
    a, b, ra, rb = synth_pair(3, length=3000, unrelated=True)
    g = csa_amd.align_batch([([a, b], [ra, rb], None, None)])[0]
    cons, strs, st = oracle_progressive([a, b], [ra, rb])
    assert g["aligned"] == strs and g["score"] == st.last_score


def test_families_vs_oracle_medium(profile_mode):
    r = rng(99)
    tasks = []
    for n, length in [(3, 900), (5, 1500), (8, 700), (16, 300)]:
        fam = random_family(r, n, length, mut=0.1, indel=0.06)
        tasks.append((fam, [r.randrange(len(f)) for f in fam], None, None))
    got = csa_amd.align_batch(tasks)
    for t, g in zip(tasks, got):
        cons, strs, st = oracle_progressive(t[0], t[1])
        assert g["status"] == 0 and g["consensus"] == cons
        assert g["aligned"] == strs
        assert g["score"] == st.last_score
        assert g["cells"] == st.cells and g["fills"] == st.fills


@pytest.mark.parametrize("length", [1, 2, 31, 63, 64, 65, 127, 128, 129, 191, 255, 256, 257, 513, 1030, 4100])
def test_cells_kernel_strip_chunk_and_block_boundaries(length, profile_mode):
    """nw_fill_cells and the direction walk (csadp_cells_tb.hip) at the boundaries of their geometry: a lane per column, 64 per
    strip, 4 strips per workgroup (wider jobs chain workgroups through HBM), 32 steps per hand-off block,
    16 steps per direction word, a 16-strip traceback window.  Families of 3..6 sequences (profile steps,
    stale borders, DeleteGappedColumns between them) against the oracle, string for string."""
    r = rng(1000 + length)
    tasks = []
    for n in (3, 4, 6):
        fam = random_family(r, n, length, mut=0.12, indel=0.05)
        fam = [f if f else b"A" for f in fam]
        tasks.append((fam, [r.randrange(len(f)) for f in fam], None, None))
    # very different lengths: few rows under many columns and the reverse
    short = bytes(r.choice(b"ACGT") for _ in range(max(1, length // 20)))
    fam = random_family(r, 3, length, mut=0.1, indel=0.03)
    tasks.append(([short] + [f if f else b"C" for f in fam], [0, 0, 0, 0], None, None))
    got = csa_amd.align_batch(tasks)
    for t, g in zip(tasks, got):
        cons, strs, st = oracle_progressive(t[0], t[1])
        assert g["status"] == 0 and g["consensus"] == cons, (length, len(t[0]))
        assert g["aligned"] == strs, (length, len(t[0]))
        assert g["score"] == st.last_score and g["fills"] == st.fills


@pytest.mark.parametrize("layout", ["fetcher", "plain"])
def test_cells_kernel_hand_off_between_workgroups_in_both_layouts(layout, monkeypatch):
    """Chunked jobs of nw_fill_cells in both layouts of its workgroups (launch_fill_cells; CSADP_CELLS_FETCH = the most workgroups of a
    launch that still get the fetcher wave): "fetcher" -- a fifth wave polls the previous chunk's granules into an LDS ring and the
    chunk's first strip reads them like every other strip; "plain" -- that strip requests its granules itself.  Widths around the
    chunk boundaries (512 columns per workgroup), one matrix of more than 256 hand-off blocks (the granules' block tag wraps) and
    more rows than a ring holds, families and pairs in one batch, against the oracle string for string."""
    monkeypatch.setenv("CSADP_BITS", "0")
    monkeypatch.setenv("CSADP_CELLS_FETCH", "256" if layout == "fetcher" else "0")
    r = rng(31337)
    tasks = []
    for length in (500, 513, 1025, 1600, 2049):
        fam = random_family(r, 3, length, mut=0.1, indel=0.04)
        tasks.append((fam, [r.randrange(len(f)) for f in fam], None, None))
    wide = random_family(r, 24, 1100, mut=0.05, indel=0.02)    # steps 22 and 23 take the 6-bit-count statement (WIDE) over three chunks
    tasks.append((wide, [r.randrange(len(f)) for f in wide], None, None))
    for ncols, nrows in ((1500, 9000), (9000, 1500), (700, 40), (40, 700)):
        base = bytes(r.choice(b"ACGT") for _ in range(max(ncols, nrows)))
        a = base[:ncols]
        b = bytes(c if r.random() > 0.1 else r.choice(b"ACGT") for c in base[:nrows])
        tasks.append(([a, b], [r.randrange(len(a)), r.randrange(len(b))], None, None))
    got = csa_amd.align_batch(tasks)
    alone = csa_amd.align_batch(tasks[6:7])[0]                 # one matrix alone: 3 workgroups, every chain on its own units
    for t, g in zip(tasks, got):
        cons, strs, st = oracle_progressive(t[0], t[1])
        assert g["status"] == 0 and g["consensus"] == cons
        assert g["aligned"] == strs and g["score"] == st.last_score
    assert alone["aligned"] == got[6]["aligned"] and alone["score"] == got[6]["score"]


def test_cells_kernel_paths_that_leave_the_traceback_window(profile_mode):
    """Profiles whose optimal path drifts far from the diagonal (a 700-letter insertion in the middle of the
    row sequence): the walk leaves its LDS window and must reload it around the current cell; the band-parallel
    walk meets bands that are crossed by hundreds of L moves (scouts give up: the band is walked exactly)."""
    r = rng(4711)
    base = bytes(r.choice(b"ACGT") for _ in range(2500))
    ins = bytes(r.choice(b"ACGT") for _ in range(700))
    fam = [base, base[:1200] + ins + base[1200:], base[:400] + base[900:], base]
    got = csa_amd.align_batch([(fam, [0, 0, 0, 0], None, None), (fam[::-1], [3, 2, 1, 0], None, None)])
    for t, g in zip([fam, fam[::-1]], got):
        pass
    for (f, rot), g in zip([(fam, [0, 0, 0, 0]), (fam[::-1], [3, 2, 1, 0])], got):
        cons, strs, st = oracle_progressive(f, rot)
        assert g["status"] == 0 and g["aligned"] == strs and g["score"] == st.last_score


def test_banded_traceback_unmerged_bands_early_ends_and_borders(monkeypatch):
    """The band-parallel walk where its shortcuts do not apply: unrelated sequences (flanking scouts rarely merge inside a
    band), a path that reaches column 0 or row 0 far from the corner (few rows under many columns and the reverse), entry
    columns right of the last start column, row counts either side of the band height, and a low-complexity profile whose
    ties send the scouts along long runs of L.  Every fill against the oracle, with the serial walk beside it."""
    r = rng(8128)
    def rand(n):
        return bytes(r.choice(b"ACGT") for _ in range(n))
    base = rand(3000)
    tasks = [
        ([rand(1500), rand(1400), rand(1600)], [0, 0, 0], None, None),                     # unrelated
        ([base[:90], base, base[100:2900]], [0, 0, 0], None, None),                         # 90 rows under 3000 columns
        ([base, base[2000:2100], base[5:]], [0, 0, 0], None, None),
        ([base[:1000] + base[1900:], base, base[:128 * 7]], [0, 0, 0], None, None),         # a 900-column run of L; 896 rows
        ([b"A" * 700 + rand(300) + b"AC" * 300, b"A" * 650 + rand(280) + b"AC" * 330, b"A" * 900 + b"AC" * 200], [0, 0, 0], None, None),
        ([base[:128 * 3 + 1], base[:128 * 3 - 1], base[:128 * 3]], [0, 0, 0], None, None),
    ]
    want = [oracle_progressive(t[0], t[1]) for t in tasks]
    # CSADP_TB_CORRIDOR: groups of 1024 start columns scouted per band around the corner-to-corner line (default 3)
    for band_min, corridor in (("1", "3"), ("1", "1"), ("1", "1000"), ("2000000000", "3")):
        monkeypatch.setenv("CSADP_TB_BAND_MIN", band_min)
        monkeypatch.setenv("CSADP_TB_CORRIDOR", corridor)
        got = csa_amd.align_batch(tasks)
        for (cons, strs, st), g in zip(want, got):
            assert g["status"] == 0 and g["consensus"] == cons
            assert g["aligned"] == strs and g["score"] == st.last_score and g["fills"] == st.fills


def test_more_banded_jobs_than_a_grid_dimension(monkeypatch):
    """The band-parallel traceback kernels take their job from the grid's y / z dimension, which ends at 65535: 66 000 tiny
    three-sequence tasks with every matrix banded (CSADP_TB_BAND_MIN=1) in one lock-step round go in two launches
    (round-3 ADVICE).  A sample against the oracle, all by their status and shape."""
    monkeypatch.setenv("CSADP_TB_BAND_MIN", "1")
    r = rng(65536)
    kinds = []
    for _ in range(40):
        fam = [bytes(r.choice(b"ACGT") for _ in range(r.randrange(2, 7))) for _ in range(3)]
        kinds.append((fam, [r.randrange(len(f)) for f in fam], None, None))
    tasks = [kinds[i % len(kinds)] for i in range(66000)]
    got = csa_amd.align_batch(tasks)
    want = [oracle_progressive(k[0], k[1]) for k in kinds]
    for i, g in enumerate(got):
        cons, strs, st = want[i % len(kinds)]
        assert g["status"] == 0 and g["consensus"] == cons and g["aligned"] == strs, i


def test_device_io_alphabet_regions_and_rotations():
    """The device-side input path (nw_pack_planes): CharAt's single wrap at every rotation/start
    combination, sub-regions, shared texts (one upload per distinct text), letters outside A,C,G,T
    inside and outside the regions (CSADP_ERR_ALPHABET only when inside, per task), empty regions
    beside full ones in one batch -- against the oracle string for string."""
    r = rng(99)
    base = [bytes(r.choice(b"ACGT") for _ in range(n)) for n in (1, 2, 63, 64, 65, 700, 2049)]
    tasks = []
    for _ in range(120):
        a, b = r.choice(base), r.choice(base)
        ra, rb = r.randrange(len(a)), r.randrange(len(b))
        sa = r.randrange(len(a) + 1) if r.random() < 0.5 else 0
        sb = r.randrange(len(b) + 1) if r.random() < 0.5 else 0
        ea = r.randrange(sa, len(a) + 1) if r.random() < 0.5 else len(a)
        eb = r.randrange(sb, len(b) + 1) if r.random() < 0.5 else len(b)
        tasks.append(([a, b], [ra, rb], [sa, sb], [ea, eb]))
    # an N inside the region of one task, outside the region of another
    dirty = bytearray(base[5]); dirty[100] = ord("N"); dirty = bytes(dirty)
    tasks.append(([dirty, base[5]], [0, 0], [0, 0], [700, 700]))          # N at rotated position 100: inside
    tasks.append(([dirty, base[5]], [0, 0], [101, 0], [700, 700]))        # region starts behind it: fine
    tasks.append(([dirty, base[5]], [650, 0], [0, 0], [100, 700]))        # rotated positions 0..99 = text 650..699, 0..49: fine
    tasks.append(([dirty, base[5]], [650, 0], [0, 0], [151, 700]))        # ... 0..100: inside
    got = csa_amd.align_batch(tasks)
    assert [g["status"] for g in got[-4:]] == [csa_amd.ERR_ALPHABET, 0, 0, csa_amd.ERR_ALPHABET]
    for t, g in zip(tasks, got):
        cons, strs, st = oracle_progressive(*t)
        if cons < 0:
            assert g["status"] == csa_amd.ERR_ALPHABET and g["aligned"] is None
            continue
        assert g["status"] == 0 and g["consensus"] == cons
        assert g["aligned"] == (strs if strs[0] is not None else None)
        if strs[0] is not None:
            assert g["score"] == st.last_score and g["progress"] == ("." if st.fills else "")


def test_errors():
    bad = csa_amd.align_batch([([b"ACGT", b"ACNT"], None, None, None), ([b"ACGT", b"ACGT"], None, None, None)])
    assert bad[0]["status"] == csa_amd.ERR_ALPHABET
    assert bad[1]["status"] == 0 and bad[1]["aligned"] == [b"ACGT", b"ACGT"]
    bad = csa_amd.align_batch([([b"ACGT", b"ACGT"], [0, 0], [0, 0], [5, 4])])
    assert bad[0]["status"] == csa_amd.ERR_ARG


def _real_cases(setname, limit=None):
    cases = [c for c in load_golden("real_pairs.json") if c["set"] == setname]
    return cases[:limit] if limit else cases


@pytest.mark.parametrize("setname", ["Primates", "Mammals"])
def test_real_pairs_golden(setname):
    """All whole-sequence pairs of the reference's example sets (config 2 + config 3):
    alignment length, SP score and FNV-1a digest of the strings must equal the values the
    compiled reference produced."""
    _, seqs = read_fasta(os.path.join(GOLDEN, "data", setname + ".txt"))
    cases = _real_cases(setname)
    tasks = [([seqs[c["a"]], seqs[c["b"]]], c["rots"], None, None) for c in cases]
    pb = csa_amd.PairBatch(tasks)
    pb.run()
    pb.sync()
    got = pb.fetch()
    pb.close()
    for c, g in zip(cases, got):
        assert g["status"] == 0
        assert g["consensus"] == c["consensus"], (c["a"], c["b"])
        assert g["score"] == c["sp"]
        assert sp_score(g["aligned"]) == c["sp"]
        assert "%08x" % fnv1a(g["aligned"]) == c["fnv1a"], (c["a"], c["b"])


def test_config2_primates_pair_vs_survey_values():
    """Config 2: Primates seq0 x seq1 with the reference's rotations 1947/1949:
    16554 x 16563 cells, len 16589, SP 15197, digest 7ee50a99."""
    _, seqs = read_fasta(os.path.join(GOLDEN, "data", "Primates.txt"))
    g = csa_amd.align_batch([([seqs[0], seqs[1]], [1947, 1949], None, None)])[0]
    assert g["consensus"] == 16589 and g["score"] == 15197
    assert "%08x" % fnv1a(g["aligned"]) == "7ee50a99"
    assert g["cells"] == 16554 * 16563


def test_full_size_synthetic_properties():
    """Config 4 shape (16 kbp synthetic circular pairs) at full size: de-gapped strings
    equal the rotated inputs, equal lengths, SP(aligned) == DP score, the score is
    invariant under exchanging the two sequences, and a repeated run is identical."""
    pairs = [synth_pair(p) for p in range(6)]
    tasks = [([a, b], [ra, rb], None, None) for a, b, ra, rb in pairs]
    swapped = [([b, a], [rb, ra], None, None) for a, b, ra, rb in pairs]
    pb = csa_amd.PairBatch(tasks + swapped)
    pb.run()
    pb.run()
    pb.sync()
    t = pb.timing()
    got = pb.fetch()
    pb.close()
    assert t["cells"] == sum(len(a) * len(b) for a, b, _, _ in pairs) * 2
    n = len(pairs)
    for i, (a, b, ra, rb) in enumerate(pairs):
        g, h = got[i], got[n + i]
        assert g["status"] == 0 and h["status"] == 0
        assert len(g["aligned"][0]) == len(g["aligned"][1]) == g["consensus"]
        assert degap(g["aligned"][0]) == rotated(a, ra)
        assert degap(g["aligned"][1]) == rotated(b, rb)
        assert sp_score(g["aligned"]) == g["score"]
        assert h["score"] == g["score"]
    again = csa_amd.align_batch(tasks[:2])
    for g, h in zip(got[:2], again):
        assert g["aligned"] == h["aligned"] and g["score"] == h["score"]


def test_full_size_pair_vs_oracle(fill_mode):
    """One 16 kbp synthetic pair, full strings against the oracle."""
    a, b, ra, rb = synth_pair(1)
    g = csa_amd.align_batch([([a, b], [ra, rb], None, None)])[0]
    cons, strs, st = oracle_progressive([a, b], [ra, rb])
    assert g["consensus"] == cons and g["aligned"] == strs and g["score"] == st.last_score


def test_large_pairs_alone_take_either_path_with_equal_results(monkeypatch):
    """At most 8 large square-ish pairs with the device to themselves are routed to nw_fill_cells and its band-parallel walk
    (FillBatch::layout, CSADP_LONE_CELLS); the bit-parallel path must give the same rows: one pair and three pairs, both routes,
    against the oracle's score (two-row fill) and string for string against each other."""
    pairs = [synth_pair(900 + i, length=n) for i, n in enumerate((5000, 9000, 6000))]
    tasks = [([a, b], [ra, rb], None, None) for a, b, ra, rb in pairs]
    routed = csa_amd.align_batch(tasks[:1]) + csa_amd.align_batch(tasks)
    monkeypatch.setenv("CSADP_LONE_CELLS", "0")
    plain = csa_amd.align_batch(tasks[:1]) + csa_amd.align_batch(tasks)
    for g, h, t in zip(routed, plain, tasks[:1] + tasks):
        assert g["status"] == 0 and h["status"] == 0
        assert g["aligned"] == h["aligned"] and g["score"] == h["score"] and g["consensus"] == h["consensus"]
        assert g["score"] == oracle_pair_score_linear(t[0], t[1])
    pb = csa_amd.PairBatch(tasks[:1])
    pb.run()
    pb.sync()
    assert pb.timing()["words_per_lane"] > 0                      # CSADP_LONE_CELLS=0: bit-parallel
    pb.close()
    monkeypatch.delenv("CSADP_LONE_CELLS")
    pb = csa_amd.PairBatch(tasks[:1])
    pb.run()
    pb.sync()
    assert pb.timing()["words_per_lane"] == 0                     # routed: the cell-per-lane kernels
    got = pb.fetch()
    pb.close()
    assert got[0]["aligned"] == routed[0]["aligned"]


def test_mixed_length_batch(fill_mode):
    """Config 5 flavour at reduced scale: lengths spanning 100x in one batch."""
    r = rng(2026)
    tasks = []
    for length in [120, 700, 2500, 9000, 300, 12000, 64, 5000]:
        fam = random_family(r, 2, length, mut=0.1, indel=0.04)
        tasks.append((fam, [r.randrange(len(f)) for f in fam], None, None))
    got = csa_amd.align_batch(tasks)
    for t, g in zip(tasks, got):
        cons, strs, st = oracle_progressive(t[0], t[1])
        assert g["aligned"] == strs and g["score"] == st.last_score


def test_long_pair_properties_100kbp():
    """Config 5 upper range: one ~100 kbp x 100 kbp pair (1e10 cells, 2.5 GB of directions; the
    reference would need 50 GB of matrices).  No CPU oracle at this size: check the
    size-independent properties -- strings re-spell the inputs, equal lengths, SP(aligned)
    equals the DP score the host re-derives along the path, swapping the sequences keeps it."""
    a, b, ra, rb = synth_pair(4242, length=100000)
    got = csa_amd.align_batch([([a, b], [ra, rb], None, None), ([b, a], [rb, ra], None, None)])
    g, h = got
    assert g["status"] == 0 and h["status"] == 0
    assert g["cells"] == len(a) * len(b)
    assert len(g["aligned"][0]) == len(g["aligned"][1]) == g["consensus"]
    assert degap(g["aligned"][0]) == rotated(a, ra) and degap(g["aligned"][1]) == rotated(b, rb)
    assert sp_score(g["aligned"]) == g["score"] == h["score"]
    assert g["score"] > 50000
    # optimality: a valid but sub-optimal path passes all of the above; the oracle's two-row fill
    # (odp_pair_score_linear, pinned to the reference in tests/test_oracle.py) knows the optimum
    assert g["score"] == oracle_pair_score_linear([a, b], [ra, rb])


def test_config5_mixed_lengths_sample():
    """A deterministic sample of config 5's length distribution (1 k .. 200 k), lengths capped
    so the CPU oracle can confirm the shorter ones string-for-string."""
    from csa_amd.synth import config5_lengths
    la, _ = config5_lengths(256)
    lengths = sorted(la)[::16][:12]            # 12 tasks spanning the distribution's lower 3/4
    tasks = []
    for i, length in enumerate(lengths):
        a, b, ra, rb = synth_pair(9000 + i, length=int(length))
        tasks.append(([a, b], [ra, rb], None, None))
    got = csa_amd.align_batch(tasks)
    for t, g in zip(tasks, got):
        assert g["status"] == 0
        assert degap(g["aligned"][0]) == rotated(t[0][0], t[1][0])
        assert degap(g["aligned"][1]) == rotated(t[0][1], t[1][1])
        assert sp_score(g["aligned"]) == g["score"]
        if len(t[0][0]) <= 6000:
            cons, strs, st = oracle_progressive(t[0], t[1])
            assert g["aligned"] == strs and g["score"] == st.last_score


@pytest.mark.parametrize("nseq", [22, 23, 24, 33, 40, 64])
def test_wide_profile_more_than_32_sequences(nseq, profile_mode):
    """Many sequences switch a round to the 6-bit-count table format (csadp_device.h): nw_fill_cells from
    i = 22 on (its byte form folds the column's gap term into the gain bytes: 12 i + 1 <= 255); 22 / 23 / 24
    sequences sit on that boundary.  64 is the reference's MAXNUMBEROFSEQS
    (csamsa.c:23)."""
    r = rng(1000 + nseq)
    fam = random_family(r, nseq, 150, mut=0.08, indel=0.05)
    rots = [r.randrange(len(f)) for f in fam]
    small = random_family(r, 3, 400)
    got = csa_amd.align_batch([(fam, rots, None, None), (small, None, None, None)])
    for task, g in zip([(fam, rots), (small, None)], got):
        cons, strs, st = oracle_progressive(task[0], task[1])
        assert g["status"] == 0 and g["consensus"] == cons
        assert g["aligned"] == strs
        assert g["score"] == st.last_score and g["fills"] == st.fills


def test_folded_gain_bytes_at_their_bounds(profile_mode):
    """nw_fill_cells' byte table holds 8 sv + 2 - leftc per letter.  Its extremes: 21 identical sequences
    already aligned (sv = i, no gaps: 12 i + 1 = 253 at the last byte-form step) and columns that are almost
    all gaps (leftc = 1, sv = 0).  Both families also cross into the 6-bit-count form at i = 22."""
    r = rng(77)
    base = bytes(r.choice(b"ACGT") for _ in range(300))
    same = [base] * 24
    # one long sequence first, then many short ones: most columns of the profile are gaps for most rows
    gappy = [bytes(r.choice(b"ACGT") for _ in range(400))] + [bytes(r.choice(b"ACGT") for _ in range(40)) for _ in range(23)]
    got = csa_amd.align_batch([(same, None, None, None), (gappy, None, None, None)])
    for fam, g in zip([same, gappy], got):
        cons, strs, st = oracle_progressive(fam, None)
        assert g["status"] == 0 and g["consensus"] == cons
        assert g["aligned"] == strs
        assert g["score"] == st.last_score and g["fills"] == st.fills


def test_adversarial_pairs_vs_oracle(fill_mode):
    """Homopolymers (steepest growth along the diagonal), all mismatches, periodic sequences (many ties), and
    partners of very different sizes in one batch.  Everything is compared with the oracle string for string."""
    r = rng(77)
    n = 5000
    rnd = bytes(r.choice(b"ACGT") for _ in range(n))
    tasks = [
        ([b"A" * n, b"A" * n], None, None, None),
        ([b"A" * n, b"C" * (n - 7)], None, None, None),
        ([b"ACGT" * (n // 4), b"CGTA" * (n // 4)], [3, 1], None, None),
        ([b"AC" * (n // 2), b"CA" * (n // 2 - 3)], None, None, None),
        ([rnd, rnd[:200]], [17, 5], None, None),
        ([rnd[:300], rnd], [0, 4000], None, None),
        ([rnd, bytes(reversed(rnd))], None, None, None),
        ([b"A" * 2000 + b"C" * 3000, b"C" * 2500 + b"A" * 2500], None, None, None),
        ([b"G", rnd], None, None, None),
    ]
    got = csa_amd.align_batch(tasks)
    for t, g in zip(tasks, got):
        cons, strs, st = oracle_progressive(t[0], t[1])
        assert g["status"] == 0 and g["consensus"] == cons
        assert g["aligned"] == strs
        assert g["score"] == st.last_score


def test_long_homopolymers(fill_mode):
    """60 kbp of the same letter: in gain form X grows by 8 per row for 60 k rows (480 k in total)."""
    n = 60000
    got = csa_amd.align_batch([([b"T" * n, b"T" * n], None, None, None),
                               ([b"T" * n, b"T" * (n - 100)], None, None, None)])
    assert got[0]["score"] == n and got[0]["aligned"] == [b"T" * n, b"T" * n]
    assert got[1]["score"] == n - 200 and got[1]["consensus"] == n
    assert got[1]["aligned"][1].count(b"-") == 100 and got[1]["aligned"][0] == b"T" * n


def test_benchmark_workload_against_reference_digests():
    """The bench's own workload: the first 32 pairs of config 4 must reproduce the compiled
    reference's strings (length, SP score, FNV-1a digest committed in config4_pairs.json)."""
    gold = load_golden("config4_pairs.json")
    pairs = [synth_pair(g["pair"]) for g in gold]
    got = csa_amd.align_batch([([a, b], [ra, rb], None, None) for a, b, ra, rb in pairs])
    for g, r in zip(gold, got):
        assert r["status"] == 0
        assert (r["consensus"], r["score"], "%08x" % fnv1a(r["aligned"])) == (g["consensus"], g["sp"], g["fnv1a"]), g["pair"]


def test_whole_config4_properties():
    """All 1024 pairs of config 4 (2.75e11 cells) in one device-resident batch, three passes:
    de-gapped rows re-spell the rotated inputs, equal lengths, SP(aligned) == DP score; ALL 1024
    results equal the compiled reference's (length, SP score, FNV-1a of the two rows:
    tests/golden/config4_all.json); a second fetch-free pass leaves the same results."""
    n = 1024
    pairs = [synth_pair(p) for p in range(n)]
    tasks = [([a, b], [ra, rb], None, None) for a, b, ra, rb in pairs]
    pb = csa_amd.PairBatch(tasks)
    for _ in range(3):
        pb.run()
    pb.sync()
    t = pb.timing()
    got = pb.fetch()
    pb.close()
    assert t["cells"] == sum(len(a) * len(b) for a, b, _, _ in pairs)
    gold = load_golden("config4_all.json")
    assert gold["pairs"] == n
    total = 0
    for p, ((a, b, ra, rb), g) in enumerate(zip(pairs, got)):
        assert g["status"] == 0
        assert len(g["aligned"][0]) == len(g["aligned"][1]) == g["consensus"]
        assert degap(g["aligned"][0]) == rotated(a, ra) and degap(g["aligned"][1]) == rotated(b, rb)
        if p % 16 == 0:
            assert sp_score(g["aligned"]) == g["score"]
        assert (g["consensus"], g["score"], "%08x" % fnv1a(g["aligned"])) == (gold["consensus"][p], gold["sp"][p], gold["fnv1a"][p]), p
        total += g["score"]
    assert total > 0


def test_config4_unrelated_variant_64_pairs():
    """SURVEY 8(d), config 4's second variant: 64 UNRELATED 16 kbp pairs (b an independent random sequence) -- the worst case
    of the negative score range and of the traceback (a gap-rich path that wanders off the corner-to-corner line).  Strings
    against the oracle for three pairs; for all 64: the size-independent properties and the score against the oracle's
    linear-space optimum (dynamicprogramming.c:990-1029, walk :1037-1047)."""
    from concurrent.futures import ThreadPoolExecutor
    n = 64
    pairs = [synth_pair(70000 + p, unrelated=True) for p in range(n)]
    tasks = [([a, b], [ra, rb], None, None) for a, b, ra, rb in pairs]
    pb = csa_amd.PairBatch(tasks)
    pb.run()
    pb.run()
    pb.sync()
    got = pb.fetch()
    pb.close()
    for (a, b, ra, rb), g in zip(pairs, got):
        assert g["status"] == 0
        assert len(g["aligned"][0]) == len(g["aligned"][1]) == g["consensus"]
        assert degap(g["aligned"][0]) == rotated(a, ra) and degap(g["aligned"][1]) == rotated(b, rb)
        assert sp_score(g["aligned"]) == g["score"]
    # all 64 records against the compiled reference's (tests/golden/unrelated_pairs.json)
    for g, r in zip(load_golden("unrelated_pairs.json"), got):
        assert (r["consensus"], r["score"], "%08x" % fnv1a(r["aligned"])) == (g["consensus"], g["sp"], g["fnv1a"]), g["pair"]
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 2)) as ex:
        want = list(ex.map(lambda t: oracle_pair_score_linear(t[0], t[1]), tasks[:16]))
    assert [g["score"] for g in got[:16]] == want
    for i in (0, 31, 63):
        cons, strs, st = oracle_progressive(tasks[i][0], tasks[i][1])
        assert got[i]["consensus"] == cons and got[i]["aligned"] == strs and got[i]["score"] == st.last_score
    # the same pairs one by one through the other entry point (a one-job batch takes the latency-shaped path)
    one = csa_amd.align_batch(tasks[:2])
    assert [o["aligned"] for o in one] == [g["aligned"] for g in got[:2]]


def test_whole_config5_in_one_batch():
    """Config 5: 256 pairs, lengths 1 k .. 200 k, 1.2e12 cells, as ONE device-resident
    batch -- possible because checkpoint mode keeps no direction planes (0.3 TB of them otherwise).
    Jobs of up to 98 strips run as chains of workgroups.  Size-independent properties on every
    result; the short ones string for string against the oracle."""
    from csa_amd.synth import config5_lengths
    la, _ = config5_lengths(256)
    tasks = []
    for i, length in enumerate(la):
        a, b, ra, rb = synth_pair(20000 + i, length=int(length))
        tasks.append(([a, b], [ra, rb], None, None))
    pb = csa_amd.PairBatch(tasks)
    pb.run()
    tm = pb.timing()
    got = pb.fetch()
    pb.close()
    assert tm["cells"] > 1.1e12 and tm["bit_parallel"] == 2 and max(la) > 190000
    checked = 0
    for t, g in zip(tasks, got):
        assert g["status"] == 0
        assert len(g["aligned"][0]) == len(g["aligned"][1]) == g["consensus"]
        assert degap(g["aligned"][0]) == rotated(t[0][0], t[1][0])
        assert degap(g["aligned"][1]) == rotated(t[0][1], t[1][1])
        assert sp_score(g["aligned"]) == g["score"]
        if len(t[0][0]) <= 2500 and checked < 12:
            cons, strs, st = oracle_progressive(t[0], t[1])
            assert g["aligned"] == strs and g["score"] == st.last_score
            checked += 1
    assert checked >= 8
    # every pair the reference can hold (both sides <= 40 000 letters: 171 of the 256; SURVEY 8d asks for full strings up to 40 k x 40 k)
    # against the compiled reference's record of it: length, SP score and the FNV-1a digest of the two aligned strings
    gold = load_golden("config5_pairs.json")
    assert len(gold) >= 170 and max(max(g["len_a"], g["len_b"]) for g in gold) > 38000
    for g in gold:
        r = got[g["index"]]
        assert (len(tasks[g["index"]][0][0]), len(tasks[g["index"]][0][1])) == (g["len_a"], g["len_b"])
        assert (r["consensus"], r["score"], "%08x" % fnv1a(r["aligned"])) == (g["consensus"], g["sp"], g["fnv1a"]), g["index"]
    # optimality of the LONG pairs (the reference cannot hold their matrices): the three longest
    # (up to 200 kbp, 4e10 cells each) and five more spread over 40 k .. 150 k, against the
    # oracle's linear-space score, on host threads (ctypes releases the GIL)
    from concurrent.futures import ThreadPoolExecutor
    order = sorted(range(len(tasks)), key=lambda i: -len(tasks[i][0][0]))
    mid = [i for i in order if 40000 <= len(tasks[i][0][0]) <= 150000]
    sample = order[:3] + mid[::max(1, len(mid) // 5)][:5]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        want = list(ex.map(lambda i: oracle_pair_score_linear(tasks[i][0], tasks[i][1]), sample))
    assert [got[i]["score"] for i in sample] == want
    assert max(len(tasks[i][0][0]) for i in sample) > 190000


def test_batches_in_flight_are_independent():
    """A streaming caller keeps several pair batches in flight on one engine (create + run + flush, fetch later).  A
    batch waits for ITS launches only (events, not the streams it shares with later batches) and its results come down
    on a copy stream: fetching the batches newest first, while older ones are still queued or running, must still
    return every batch's own results.  Small batches (side by side on rotating streams) and batches that fill the
    chip (first in, first out on one stream) take different paths through the engine."""
    r = rng(4242)
    for npairs, length in ((6, 900), (300, 700)):
        sets = []
        for b in range(4):
            tasks = []
            for i in range(npairs):
                a, c, ra, rc = synth_pair(100000 + 1000 * b + i + 17 * length, length=length)
                tasks.append(([a, c], [ra, rc], None, None))
            sets.append(tasks)
        batches = []
        for tasks in sets:
            pb = csa_amd.PairBatch(tasks)
            pb.run()
            pb.flush()
            batches.append(pb)
        for b in (3, 1, 2, 0):                      # newest first, then out of order
            got = batches[b].fetch()
            for t, g in zip(sets[b][:8] + sets[b][-2:], got[:8] + got[-2:]):
                cons, strs, st = oracle_progressive(t[0], t[1])
                assert g["status"] == 0 and g["aligned"] == strs and g["score"] == st.last_score
            for t, g in zip(sets[b], got):          # every pair: the rows spell the rotated inputs, SP = score
                assert degap(g["aligned"][0]) == rotated(t[0][0], t[1][0])
                assert degap(g["aligned"][1]) == rotated(t[0][1], t[1][1])
                assert sp_score(g["aligned"]) == g["score"]
        for pb in batches:
            pb.close()


def test_score_range_limit_of_the_gain_form(monkeypatch):
    """csadp_engine.cpp, layout_cells: the 32-bit kernel keeps X = 4 H + 4 i r with two tag bits, so a fill needs
    i (2 nrows + ncols) 4 + 64 < 2^31, else CSADP_ERR_RANGE for the batch (the reference's counterpart is an int that would
    overflow silently, dynamicprogramming.c:996-1012).  A real fill at that bound is 180 M letters against one sequence or 2.8 M
    against 63: the test lowers the bound (CSADP_TEST_RANGE_LOG2) until the LAST step of a 5-sequence family crosses it --
    the steps before it run, the call reports the range error, and the same task passes again at the real bound."""
    r = rng(4711)
    fam = random_family(r, 5, 3000, mut=0.1, indel=0.03)
    small = random_family(r, 3, 300)
    cons, strs, st = oracle_progressive(fam, None)
    need = [i * (2 * len(sorted(fam, key=len)[i]) + cons) * 4 + 64 for i in range(1, 5)]      # roughly: consensus grows from step to step
    bits = max(need).bit_length() - 1                     # 2^bits <= the largest step's need
    assert (1 << bits) > need[1]                          # ... and the early steps fit
    monkeypatch.setenv("CSADP_TEST_RANGE_LOG2", str(bits))
    with pytest.raises(csa_amd.CsadpError) as e:
        csa_amd.align_batch([(fam, None, None, None), (small, None, None, None)])
    assert e.value.code == csa_amd.ERR_RANGE
    got = csa_amd.align_batch([(small, None, None, None)])            # the library is in order after the error
    assert got[0]["status"] == 0 and got[0]["aligned"] == oracle_progressive(small, None)[1]
    monkeypatch.delenv("CSADP_TEST_RANGE_LOG2")
    got = csa_amd.align_batch([(fam, None, None, None)])
    assert got[0]["status"] == 0 and got[0]["aligned"] == strs and got[0]["score"] == st.last_score


def test_batch_that_does_not_fit_the_device_memory(monkeypatch, capfd):
    """csadp_engine.cpp, alloc_buffers: a batch whose arena does not fit the device's free memory (less 256 MB) is refused with
    CSADP_ERR_RANGE and a line on stderr -- the reference's counterpart is an unchecked malloc per matrix row
    (dynamicprogramming.c:964-981).  CSADP_TEST_HBM_LIMIT_MB pretends a device with 264 MB free (8 MB usable): a 3-sequence family of 12 000
    letters needs 36 MB of direction words and more per step, a family of 400 fits."""
    r = rng(1812)
    big = random_family(r, 3, 12000, mut=0.1, indel=0.02)
    small = random_family(r, 3, 400)
    csa_amd.shutdown()                                    # drop the cached arenas of earlier tests: the batch must ask for memory
    csa_amd.init(device=0)
    monkeypatch.setenv("CSADP_TEST_HBM_LIMIT_MB", "264")
    got = csa_amd.align_batch([(small, None, None, None)])
    assert got[0]["status"] == 0 and got[0]["aligned"] == oracle_progressive(small, None)[1]
    with pytest.raises(csa_amd.CsadpError) as e:
        csa_amd.align_batch([(big, None, None, None)])
    assert e.value.code == csa_amd.ERR_RANGE
    assert "GiB of HBM" in capfd.readouterr().err
    monkeypatch.delenv("CSADP_TEST_HBM_LIMIT_MB")
    got = csa_amd.align_batch([(big, None, None, None)])
    cons, strs, st = oracle_progressive(big, None)
    assert got[0]["status"] == 0 and got[0]["aligned"] == strs and got[0]["score"] == st.last_score
