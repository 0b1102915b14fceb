"""GPU tests (-m gpu) aimed at the bit-parallel kernels (csadp_bits.hip): word / strip / block
boundaries of the 32-column words, the hand-off between the strips of a workgroup (LDS) and between the
chunks of a matrix (granules in HBM), the checkpoint + replay traceback -- with 1, 2 and 4 words per lane.
Expected strings come from the oracle (oracle/csa_dp_oracle.c) on the same inputs."""
import pytest

import csa_amd
from helpers import degap, oracle_progressive, random_family, rng, rotated, sp_score

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def device():
    csa_amd.init(device=0)
    yield


@pytest.fixture(autouse=True)
def bit_parallel_path_also_for_large_pairs_alone(monkeypatch):
    """This module tests the bit-parallel kernels: a few large pairs alone would otherwise take the cell-per-lane path (FillBatch::layout)."""
    monkeypatch.setenv("CSADP_LONE_CELLS", "0")


# words of 32 columns per lane: the engine picks 1, 2 or 3 by the shape of the batch (layout_bits); every test below runs with
# each of the three kernels
@pytest.fixture(params=["1 word", "2 words", "3 words", "4 words"])
def bits_mode(request, monkeypatch):
    monkeypatch.setenv("CSADP_BITS_WORDS", request.param.split()[0])
    return request.param


def related(r, n, m, sub=0.1, indel=0.02):
    a = bytes(r.choice(b"ACGT") for _ in range(n))
    out = bytearray()
    for ch in a:
        x = r.random()
        if x < indel:
            continue
        if x < 2 * indel:
            out.append(r.choice(b"ACGT"))
        out.append(r.choice(b"ACGT") if r.random() < sub else ch)
    b = bytes(out[:m]) if m is not None else bytes(out)
    return a, (b or b"A")


def check(tasks):
    got = csa_amd.align_batch(tasks)
    for t, g in zip(tasks, got):
        cons, strs, st = oracle_progressive(t[0], t[1])
        assert g["status"] == 0 and g["consensus"] == cons, (len(t[0][0]), len(t[0][1]))
        assert g["aligned"] == strs, (len(t[0][0]), len(t[0][1]))
        assert g["score"] == st.last_score


def test_word_strip_and_block_boundaries(bits_mode):
    """Lengths around 32 (word), 2048 (strip = wave) and 32 / 64 rows (hand-off block, lane skew).
    The shorter sequence is the profile (columns), the longer one the rows."""
    r = rng(101)
    tasks = []
    for cols in (1, 31, 32, 33, 63, 64, 65, 2047, 2048, 2049, 4095, 4096, 4097):
        for extra in (0, 1, 31, 33, 95):
            a, b = related(r, cols + extra, cols)
            if len(b) < cols:
                b = b + bytes(r.choice(b"ACGT") for _ in range(cols - len(b)))
            tasks.append(([a, b], None, None, None))
    check(tasks)


def test_many_strips_few_rows_and_few_strips_many_rows(bits_mode):
    r = rng(102)
    wide = bytes(r.choice(b"ACGT") for _ in range(9000))
    tasks = [([wide[100:100 + n], wide], None, None, None) for n in (1, 2, 30, 33, 64, 65, 130)]
    # equal lengths: the reference keeps input order, so the first sequence is the profile
    tall = bytes(r.choice(b"ACGT") for _ in range(7000))
    tasks += [([tall[:40], tall[:40]], None, None, None)]
    check(tasks)


def test_sixteen_strips_and_beyond(bits_mode):
    """32768 columns = 16 waves of one word per lane in one workgroup.  Wider jobs: a workgroup per chunk of
    strips, chained through epoch-tagged granules in HBM."""
    r = rng(103)

    def wide_task(cols):
        a, b = related(r, cols + 40, None, sub=0.05, indel=0.005)
        b = (b + bytes(r.choice(b"ACGT") for _ in range(cols)))[:cols]
        rows = a[:600] + a[-600:]                       # 1200 rows against a wide profile: large drift
        return ([rows, b], None, None, None)

    for cols in (32768, 32800, 100003):
        check([wide_task(cols)])
    # 32 jobs, two of them wider than one workgroup
    small = []
    for n in range(30):
        a, b = related(r, 300 + 37 * n, None)
        small.append(([a, b], None, None, None))
    check([wide_task(70001)] + small + [wide_task(33000)])


def test_paths_that_leave_the_diagonal(bits_mode):
    """Long horizontal and vertical stretches (the replay must follow the path through many
    blocks of one strip, and across strip boundaries), borders reached early."""
    r = rng(104)
    core = bytes(r.choice(b"ACGT") for _ in range(3000))
    junk = bytes(r.choice(b"AC") for _ in range(2500))
    tasks = [
        ([core, junk + core], None, None, None),            # 2500 leading columns unmatched
        ([core, core + junk], None, None, None),
        ([junk[:700] + core, core], None, None, None),
        ([core[:1500] + junk + core[1500:], core], None, None, None),   # a 2500-row insertion in the middle
        ([b"G" * 2100, b"T" * 2300], None, None, None),     # nothing matches
        ([b"GT" * 1100, b"TG" * 1200], None, None, None),   # everything ties
        ([core, core], [17, 17], None, None),
    ]
    check(tasks)


def test_batch_of_mixed_sizes_in_one_launch(bits_mode):
    """Jobs with 1..5 strips share a launch (block size = the largest job)."""
    r = rng(105)
    tasks = []
    for n in (50, 700, 2048, 3000, 5000, 9000, 300, 2049):
        a, b = related(r, n, None)
        tasks.append(([a, b], [r.randrange(len(a)), r.randrange(len(b))], None, None))
    check(tasks)


def test_pair_batch_pipelined_passes_agree(bits_mode):
    """The device-resident batch API with several merged passes in flight: every pass must leave
    the same results (slots are independent)."""
    r = rng(106)
    tasks = []
    for n in (1500, 2500, 4100):
        a, b = related(r, n, None)
        tasks.append(([a, b], None, None, None))
    pb = csa_amd.PairBatch(tasks)
    for _ in range(11):
        pb.run()
    pb.sync()
    got = pb.fetch()
    pb.close()
    for t, g in zip(tasks, got):
        cons, strs, st = oracle_progressive(t[0], t[1])
        assert g["aligned"] == strs and g["score"] == st.last_score


@pytest.mark.parametrize("words", ["1", "2", "3", "4"])
def test_jobs_of_one_and_two_strips_share_workgroups(words, monkeypatch):
    """Jobs narrower than four strips share four-wave workgroups of nw_fill_bits (first fit over a table of {job, strip} per wave;
    csadp_bits.hip, PACK; csadp_engine.cpp, layout_bits): workgroups with empty places, one- to four-strip jobs side by side, jobs
    whose rows end long before their neighbour's, a batch of one-strip jobs only -- pipelined (several passes per launch) and alone,
    shared and one workgroup per job (CSADP_BITS_PACK=0): the same rows, the oracle's."""
    monkeypatch.setenv("CSADP_BITS_WORDS", words)
    r = rng(2200 + int(words))
    w = int(words)
    one, two = 2048 * w, 4096 * w                        # columns of one / two strips at this many words per lane
    lens = [one - 7, two - 5, 3, one + 1, two, one, 60, one // 2, two - 1, one + 300, 900]     # 11 jobs of one and two strips
    if w <= 2:                                           # ... and of three and four (first fit: 4 | 3 + 1 | 2 + 2 | 2 + 1 + 1 | ...)
        lens += [2 * two - 3, two + one - 11, two + 1, 2 * two, two + one]
    tasks = []
    for n in lens:
        a, b = related(r, n, n)                          # neither sequence longer than n: whichever becomes the columns fits the strips meant
        tasks.append(([a, b], [r.randrange(len(a)), r.randrange(len(b))], None, None))
    only_one = [t for t, n in zip(tasks, lens) if n <= one][:7]                               # 7 one-strip jobs: workgroups of four, the last with three
    want = [oracle_progressive(t[0], t[1]) for t in tasks]
    for batch, exp in ((tasks, want), (only_one, [want[tasks.index(t)] for t in only_one])):
        rows = None
        for pack in ("1", "0"):
            monkeypatch.setenv("CSADP_BITS_PACK", pack)
            pb = csa_amd.PairBatch(batch)
            for passes in (1, 9):
                for _ in range(passes):
                    pb.run()
                pb.sync()
                got = pb.fetch()
                for g, (cons, strs, st) in zip(got, exp):
                    assert g["status"] == 0 and g["aligned"] == strs and g["score"] == st.last_score and g["consensus"] == cons, (pack, passes)
                assert rows is None or rows == [g["aligned"] for g in got]
                rows = [g["aligned"] for g in got]
            pb.close()


def test_pair_batch_run_once_then_many_times(bits_mode):
    """A pipelined batch of one workgroup per job whose single pass does not fill the chip keeps a second shape for a pass
    flushed alone onto an idle device (one word per lane, four strips per workgroup over all compute units:
    csadp_engine.h).  Run once, fetch; run many times, fetch; alone again: the same results each time, whichever shape
    took the pass -- the timing record says which."""
    r = rng(1106)
    base = bytes(r.choice(b"ACGT") for _ in range(12000))
    tasks = []
    for i in range(130):
        n = (2100, 4200, 6300, 8000, 5000)[i % 5] + i
        o = r.randrange(len(base) - n)
        a = base[o:o + n]
        b = bytearray(a)
        for _ in range(n // 9):
            b[r.randrange(len(b))] = r.choice(b"ACGT")
        for _ in range(n // 70):
            q = r.randrange(len(b) - 8)
            if r.random() < 0.5:
                del b[q:q + 1 + i % 4]
            else:
                b[q:q] = bytes(r.choice(b"ACGT") for _ in range(1 + i % 3))
        tasks.append(([a, bytes(b)], [r.randrange(len(a)), r.randrange(len(b))], None, None))
    sample = (0, 1, 2, 3, 4, 77, 129)
    want = {i: oracle_progressive(tasks[i][0], tasks[i][1]) for i in sample}
    words = int(bits_mode.split()[0])
    pb = csa_amd.PairBatch(tasks)
    seen, first = [], None
    for passes in (1, 7, 1, 2, 1):
        for _ in range(passes):
            pb.run()
        pb.sync()
        seen.append(pb.timing()["words_per_lane"])
        got = pb.fetch()
        _properties(tasks, got)
        for i in sample:
            cons, strs, st = want[i]
            assert got[i]["aligned"] == strs and got[i]["score"] == st.last_score and got[i]["consensus"] == cons, (passes, i)
        rows = [g["aligned"] for g in got]
        assert first is None or rows == first, passes
        first = rows
    pb.close()
    assert seen[1] == words and seen[3] == words, seen
    assert seen[0] == seen[2] == seen[4] == 1, seen              # the lone passes took the spread shape (one word per lane)


def test_first_step_of_families_uses_unit_borders_only():
    """Progressive tasks: step 1 runs bit-parallel, later steps the profile kernel; sub-regions and
    rotations included.  (Golden families cover this too; this one adds longer sequences.)"""
    r = rng(107)
    tasks = []
    for n in (3, 5):
        fam = random_family(r, n, 2600, mut=0.08, indel=0.03)
        tasks.append((fam, [r.randrange(len(f)) for f in fam], None, None))
    got = csa_amd.align_batch(tasks)
    for t, g in zip(tasks, got):
        cons, strs, st = oracle_progressive(t[0], t[1])
        assert g["status"] == 0 and g["aligned"] == strs
        for s, row in zip(t[0], g["aligned"]):
            assert degap(row) in (rotated(s, t[1][t[0].index(s)]),)


def _properties(tasks, got):
    for t, g in zip(tasks, got):
        assert g["status"] == 0
        rots = t[1] or [0, 0]
        assert degap(g["aligned"][0]) == rotated(t[0][0], rots[0])
        assert degap(g["aligned"][1]) == rotated(t[0][1], rots[1])
        assert len(g["aligned"][0]) == len(g["aligned"][1]) == g["consensus"]
        assert g["score"] == sp_score(g["aligned"])


def test_thousands_of_small_jobs_in_one_batch(bits_mode):
    """4096 pairs of ~400 letters: one single-wave workgroup each, 4096 replay workgroups."""
    r = rng(108)
    base = bytes(r.choice(b"ACGT") for _ in range(6000))
    tasks = []
    for i in range(4096):
        o = r.randrange(5000)
        a = base[o:o + 350 + i % 97]
        b = bytearray(a)
        for _ in range(len(a) // 12):
            b[r.randrange(len(b))] = r.choice(b"ACGT")
        del b[10:10 + i % 7]
        tasks.append(([a, bytes(b)], None, None, None))
    pb = csa_amd.PairBatch(tasks)
    for _ in range(5):
        pb.run()
    got = pb.fetch()
    pb.close()
    _properties(tasks, got)
    cons, strs, st = oracle_progressive(tasks[77][0], None)
    assert got[77]["aligned"] == strs and got[77]["score"] == st.last_score


def test_chip_filling_batch_of_two_to_five_strip_jobs(bits_mode):
    """700 pairs of 2-5 strips (at one word per lane) = ~2400 strips per launch, several waves on every SIMD; every
    result by its properties, a sample of them (every strip count, the longest) against the oracle."""
    r = rng(1109)
    base = bytes(r.choice(b"ACGT") for _ in range(20000))
    lens = (2100, 4200, 6300, 8400, 2049, 4097, 6145, 8193, 3000, 7000)
    tasks = []
    for i in range(700):
        n = lens[i % len(lens)] + (i % 13)
        o = r.randrange(len(base) - n)
        a = base[o:o + n]
        b = bytearray(a)
        for _ in range(n // 11):
            b[r.randrange(len(b))] = r.choice(b"ACGT")
        for _ in range(n // 60):
            q = r.randrange(len(b) - 8)
            if r.random() < 0.5:
                del b[q:q + 1 + i % 5]
            else:
                b[q:q] = bytes(r.choice(b"ACGT") for _ in range(1 + i % 4))
        tasks.append(([a, bytes(b)], [r.randrange(len(a)), r.randrange(len(b))], None, None))
    pb = csa_amd.PairBatch(tasks)
    for _ in range(3):
        pb.run()
    pb.sync()
    tm = pb.timing()
    assert tm["bit_parallel"] == 2 and tm["words_per_lane"] == int(bits_mode.split()[0])
    got = pb.fetch()
    pb.close()
    _properties(tasks, got)
    for i in (0, 1, 2, 3, 4, 5, 6, 7, 13, 697):
        cons, strs, st = oracle_progressive(tasks[i][0], tasks[i][1])
        assert got[i]["aligned"] == strs and got[i]["score"] == st.last_score, i


def test_extreme_aspect_ratios(bits_mode):
    r = rng(109)
    long = bytes(r.choice(b"ACGT") for _ in range(150000))
    tasks = [
        ([long, long[70000:70016]], None, None, None),      # 16 columns x 150 k rows: 4690 blocks, one strip
        ([long[:9], long[:120000]], None, None, None),      # 9 columns again (the shorter one is the profile)
        ([long[:1], long[5:6]], None, None, None),
        ([long[:40000], long[100:40100]], None, None, None),   # 40 k x 40 k: 20 strips of one word per lane (1 wide job)
    ]
    got = csa_amd.align_batch(tasks)
    _properties(tasks, got)


def test_gap_runs_repeats_and_unrelated_cores(bits_mode):
    """Paths that leave the traceback's windows and diagonals: a 2500-letter overhang in front of a shared core (one
    long gap run), shifted dinucleotide repeats (a gap move every other cell), homopolymers of two different letters."""
    r = rng(110)
    tasks = []
    for n in (40, 130, 200, 700, 2048, 2100, 5000, 9000):
        a, b = related(r, n, None)
        tasks.append(([a, b], [r.randrange(len(a)), r.randrange(len(b))], None, None))
    core = bytes(r.choice(b"ACGT") for _ in range(3000))
    junk = bytes(r.choice(b"AC") for _ in range(2500))
    tasks += [([core, junk + core], None, None, None), ([b"GT" * 1100, b"TG" * 1200], None, None, None),
              ([b"G" * 2100, b"T" * 2300], None, None, None)]
    check(tasks)
    pb = csa_amd.PairBatch(tasks)
    for _ in range(5):
        pb.run()
    got = pb.fetch()
    pb.close()
    assert all(g["status"] == 0 for g in got)


def test_traceback_leaves_its_windows_and_diagonals(bits_mode):
    """The windowed traceback keeps 256 columns around a planned line per 32-step piece and follows five diagonals per
    iteration: blocks of 1-700 inserted / deleted letters (the path leaves the windows: planned afresh), runs of three
    and more gap moves one way (a sixth diagonal), indels every few letters in alternating directions (the path
    oscillates between diagonals), lengths 3 : 4 (the plan's slope), and all of it across strip boundaries."""
    r = rng(4242)
    base = bytes(r.choice(b"ACGT") for _ in range(13000))

    def edited(src, blocks):
        out = bytearray(src)
        for n in blocks:
            q = r.randrange(len(out) - n - 1)
            if r.random() < 0.5:
                del out[q:q + n]
            else:
                out[q:q] = bytes(r.choice(b"ACGT") for _ in range(n))
        return bytes(out)

    def rippled(src, period, run):
        out = bytearray()
        for i, ch in enumerate(src):
            ph = (i // period) % 2
            if i % period < run and ph == 0:
                continue                                    # `run` letters deleted ...
            out.append(ch)
            if i % period < run and ph == 1:
                out.append(r.choice(b"ACGT"))               # ... then `run` inserted, one period on
        return bytes(out)

    tasks = [
        ([base, edited(base, [1, 2, 3, 5, 40, 130, 300, 700])], None, None, None),
        ([base[:7000], edited(base[:7000], [260, 270, 280, 512, 33])], [3000, 100], None, None),
        ([base[:9000], base], None, None, None),                       # 9000 columns x 13000 rows
        ([base[2000:11000], edited(base, [64, 65])], None, None, None),
        ([base[:6000], rippled(base[:6000], 7, 1)], None, None, None),
        ([base[:6000], rippled(base[:6000], 9, 3)], None, None, None),
        ([base[:6000], rippled(base[:6000], 16, 5)], [17, 4000], None, None),
        ([base[:2500], bytes(r.choice(b"ACGT") for _ in range(2600))], None, None, None),
    ]
    check(tasks)


def test_chunked_fills_either_side_of_the_epoch_wrap():
    """The granules between the workgroups of a chunked fill (bit-parallel and cell-per-lane) are valid when they carry their
    launch's epoch, 24 bits of a process-wide counter that skips 0 -- zeroed granules must never look valid (round-2 ADVICE: a
    21-bit tag that could be 0 let a consumer take unwritten words).  Six chunked passes with the counter set just below its
    wrap: epochs 0xfffffe, 0xffffff, 1, 2, ... -- every result the oracle's."""
    r = rng(707)
    a, b = related(r, 9000, None)                      # one pair, 5 strips: two chunks of four (one wave per SIMD)
    fam = [related(r, 900, None)[1] for _ in range(3)]   # profile steps of 8 strips: two chunks of the cell-per-lane kernel
    want_pair = oracle_progressive([a, b], [5, 3])
    want_fam = oracle_progressive(fam, None)
    old = csa_amd.debug_set_epoch(0xfffffd)
    try:
        for _ in range(3):
            g = csa_amd.align_batch([([a, b], [5, 3], None, None)])[0]
            assert g["status"] == 0 and g["aligned"] == want_pair[1] and g["score"] == want_pair[2].last_score
            g = csa_amd.align_batch([(fam, None, None, None)])[0]
            assert g["status"] == 0 and g["aligned"] == want_fam[1]
        assert csa_amd.debug_set_epoch(0) > 0xffffff       # the raw counter went past 24 bits: the epochs wrapped, skipping 0
    finally:
        csa_amd.debug_set_epoch(max(old, 16))


def test_abort_word_triggers_the_chunk_by_chunk_repeat(monkeypatch, capfd):
    """The chunked fills (nw_fill_bits_wide, nw_fill_cells) wait across workgroups with bounded spins; when
    one runs out the batch's abort word is raised and the host repeats the pass chunk by chunk -- one launch
    per chunk index, every producer finished before its consumer starts.  CSADP_TEST_FORCE_ABORT raises the
    word behind a normal launch: the repeat path must run (the library says so on stderr) and the results
    must be the oracle's.  Covers a pair wider than one chunk (bit-parallel, 4-strip chunks) and families
    whose profile steps span several chunks of the cell-per-lane kernel."""
    r = rng(606)
    a, b = related(r, 9000, None)
    tasks = [([a, b], [11, 7], None, None)]
    fam = [related(r, 700, None)[1] for _ in range(4)]
    tasks.append((fam, [0, 1, 2, 3], None, None))
    want = [oracle_progressive(t[0], t[1]) for t in tasks]
    monkeypatch.setenv("CSADP_TEST_FORCE_ABORT", "1")
    got = [csa_amd.align_batch([t])[0] for t in tasks]
    err = capfd.readouterr().err
    assert "repeating the pass chunk by chunk" in err
    for g, (cons, strs, st) in zip(got, want):
        assert g["status"] == 0 and g["consensus"] == cons and g["aligned"] == strs and g["score"] == st.last_score


@pytest.mark.parametrize("forced", [False, True])
def test_chunked_launch_with_shared_workgroups(forced, monkeypatch, capfd):
    """A batch with jobs wider than four strips is a chunked launch; the last, partial chunk of a job and the jobs narrower than a chunk share
    workgroups there too (a table of {job, strip} per wave; csadp_engine.cpp, layout_bits): at one word per lane 13 pairs of one to ten strips
    -- tails of one, two and three strips, whole narrow jobs, a job of exactly two chunks -- pipelined and alone, shared and one workgroup per
    chunk (CSADP_BITS_PACK=0), and once more with the abort word raised behind every launch (the repeat path takes the table level by level):
    the same rows, the oracle's."""
    monkeypatch.setenv("CSADP_BITS_WORDS", "1")
    r = rng(3300)
    strip = 2048
    lens = [5 * strip - 9, 900, 6 * strip + 1, 2 * strip, 9 * strip + 700, 3 * strip - 1, 8 * strip, 40, 7 * strip - 300, strip + 5, 4 * strip + 1, 10 * strip - 3, 3000]
    tasks = []
    for n in lens:
        a, b = related(r, n, n)
        tasks.append(([a, b], [r.randrange(len(a)), r.randrange(len(b))], None, None))
    want = [oracle_progressive(t[0], t[1]) for t in tasks]
    if forced:
        monkeypatch.setenv("CSADP_TEST_FORCE_ABORT", "1")
    rows = None
    for pack in ("1", "0"):
        monkeypatch.setenv("CSADP_BITS_PACK", pack)
        pb = csa_amd.PairBatch(tasks)
        for passes in (1, 5):
            for _ in range(passes):
                pb.run()
            pb.sync()
            got = pb.fetch()
            for g, (cons, strs, st) in zip(got, want):
                assert g["status"] == 0 and g["aligned"] == strs and g["score"] == st.last_score and g["consensus"] == cons, (pack, passes)
            assert rows is None or rows == [g["aligned"] for g in got]
            rows = [g["aligned"] for g in got]
        pb.close()
    if forced:
        assert "repeating the pass chunk by chunk" in capfd.readouterr().err


@pytest.mark.parametrize("slow", [1, 4])
def test_publisher_wave_that_falls_behind_its_strip(slow, monkeypatch):
    """Helper-wave layout of nw_fill_cells (launches of at most 256 workgroups): a chunk's last strip hands its values to the
    publisher wave through an 8-block LDS ring and may only overwrite a block's slots once the publisher has sent BOTH halves of
    it (`taken` >= block + 1; csadp_cells.hip, run_strip).  Round-4 ADVICE: the strip waited for one block less, and not at
    all at block 8, so a publisher seven blocks behind -- a preempted or shared device -- sent overwritten values under a valid
    tag: a wrong alignment, silently.  CSADP_TEST_SLOW_PUBLISHER makes the publisher sleep `slow` x 8 000 cycles per half block
    (a strip's block takes ~3 000): it falls behind at once, and only the strip's wait keeps the hand-off right.  Matrices of
    several chunks and far more than 8 blocks: a 3-sequence family (profile step) and a pair on the cell-per-lane path."""
    r = rng(909 + slow)
    fam = [related(r, 2600, None)[1] for _ in range(3)]
    a, b = related(r, 5000, None)
    tasks = [(fam, [5, 0, 9], None, None)]
    want = [oracle_progressive(t[0], t[1]) for t in tasks]
    wantp = oracle_progressive([a, b], [3, 1])
    monkeypatch.setenv("CSADP_TEST_SLOW_PUBLISHER", str(slow))
    got = csa_amd.align_batch(tasks)
    for g, (cons, strs, st) in zip(got, want):
        assert g["status"] == 0 and g["consensus"] == cons and g["aligned"] == strs and g["score"] == st.last_score
    monkeypatch.setenv("CSADP_BITS", "0")                     # the pair's first fill on nw_fill_cells too
    g = csa_amd.align_batch([([a, b], [3, 1], None, None)])[0]
    assert g["status"] == 0 and g["aligned"] == wantp[1] and g["score"] == wantp[2].last_score


@pytest.mark.parametrize("order", ["0", "1", None])
def test_work_list_order_of_a_profile_launch(order, monkeypatch):
    """nw_fill_cells takes its workgroups from a list: job by job (CSADP_CELLS_ORDER=0: launches the device holds at once) or chunk level
    by chunk level (=1: larger launches, so that a resident workgroup's producer is through, not still on its way; csadp_engine.cpp,
    layout_cells).  Either way a job's chunks are in ascending order and the results are the oracle's: 160 families of 3 sequences of
    1 500-2 600 letters (12-21 strips: 3-6 chunks per matrix; four round groups of ~160 workgroups, together more than the device holds),
    forced to one order, to the other, and left to the rule."""
    if order is not None:
        monkeypatch.setenv("CSADP_CELLS_ORDER", order)
    r = rng(4242)
    tasks = []
    for i in range(160):
        fam = random_family(r, 3, 1500 + 275 * (i % 5), mut=0.08, indel=0.03)
        tasks.append((fam, [r.randrange(len(f)) for f in fam], None, None))
    before = csa_amd.recoveries()
    got = csa_amd.align_batch(tasks)
    assert csa_amd.recoveries() == before
    for t, g in zip(tasks, got):
        assert g["status"] == 0 and len(set(len(x) for x in g["aligned"])) == 1
    for i in range(0, 160, 7):
        cons, strs, st = oracle_progressive(tasks[i][0], tasks[i][1])
        assert got[i]["consensus"] == cons and got[i]["aligned"] == strs and got[i]["score"] == st.last_score


@pytest.mark.parametrize("walk", ["serial", "banded"])
def test_thousands_of_small_families_in_one_batch(walk, monkeypatch):
    """The N-sequence twin of the test above: 2048 families of 3-6 sequences of 60-700 letters through csadp_align_batch -- lock-step
    rounds of up to 2048 profile fills (nw_fill_cells: one- to six-strip matrices, thousands of workgroups per launch: the plain layout,
    two workgroups per compute unit), serial walks, trace application and DeleteGappedColumns for every task between two rounds.  Every
    result by its properties (equal lengths, rows re-spell the rotated regions); 64 of them string for string against the oracle."""
    if walk == "banded":
        monkeypatch.setenv("CSADP_TB_BAND_MIN", "1")         # every matrix takes the band-parallel walk (scout / resolve / emit / gather)
    r = rng(2048)
    tasks = []
    for i in range(2048):
        fam = random_family(r, 3 + i % 4, r.choice([60, 130, 300, 700]), mut=0.1, indel=0.05)
        fam = [f if f else b"T" for f in fam]
        tasks.append((fam, [r.randrange(len(f)) for f in fam], None, None))
    got = csa_amd.align_batch(tasks)
    for t, g in zip(tasks, got):
        assert g["status"] == 0 and len(g["aligned"]) == len(t[0])
        assert len(set(len(x) for x in g["aligned"])) == 1 and len(g["aligned"][0]) == g["consensus"]
        for s, row in enumerate(g["aligned"]):
            assert degap(row) == rotated(t[0][s], t[1][s])
    for i in range(0, 2048, 32):
        cons, strs, st = oracle_progressive(tasks[i][0], tasks[i][1])
        assert got[i]["consensus"] == cons and got[i]["aligned"] == strs and got[i]["fills"] == st.fills


def test_many_task_batches_run_four_round_groups_without_recoveries():
    """From 128 tasks on csadp_align_batch drives FOUR round groups side by side (csadp_api.cpp): up to four chunked fills of nw_fill_cells share
    the chip, each waiting across workgroups for its own producers.  The bounded waits must not run out in such batches (a run-out is repaired --
    the pass is repeated chunk by chunk -- but costs half a second): families whose matrices span 3 to 24 workgroups, twice; recoveries unchanged;
    a sample against the oracle."""
    r = rng(4128)
    tasks = []
    for i in range(160):
        fam = random_family(r, r.choice([3, 5, 8]), r.choice([1500, 3000, 6000]) if i % 8 else 12000, mut=0.08, indel=0.02)
        tasks.append((fam, None, None, None))
    before = csa_amd.recoveries()
    for _ in range(2):
        got = csa_amd.align_batch(tasks)
        ph = csa_amd.last_batch_phases()
        assert ph["round_groups"] == 4 and all(g["status"] == 0 for g in got)
    assert csa_amd.recoveries() == before
    for i in (0, 8, 77, 159):
        cons, strs, st = oracle_progressive(tasks[i][0], None)
        assert got[i]["consensus"] == cons and got[i]["aligned"] == strs and got[i]["score"] == st.last_score
