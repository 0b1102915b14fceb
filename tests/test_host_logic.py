"""CPU tests of the PRODUCT's host side: C-ABI surface, error behaviour without a device,
the progressive host logic (through the csadp_debug seam, with an oracle-backed matrix
filler supplied by the test), partitioning, FASTA loader, workload generator."""
import ctypes
import os
import subprocess

import pytest

import csa_amd
from csa_amd.synth import config5_lengths, synth_pair
from helpers import (GOLDEN, ROOT, golden_aligned, golden_task, load_golden, oracle_filler, oracle_progressive,
                     random_family, read_fasta, rng, rotated, degap)


def test_library_exports_every_declared_symbol():
    lib = csa_amd.lib()
    for name in csa_amd.EXPORTS:
        assert hasattr(lib, name), name
    # every function declared in the public headers is in EXPORTS
    import re
    declared = set()
    for hdr in ("csadp.h", "csadp_debug.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        declared |= set(re.findall(r"\b(csadp_[a-z_0-9]+)\s*\(", text))
    declared -= {"csadp_debug_fill_fn"}
    assert declared == set(csa_amd.EXPORTS)
    assert lib.csadp_version() == 500


def test_dropin_object_defines_progressivedp():
    obj = os.path.join(ROOT, "csa_amd", "csadp_dropin.o")
    assert os.path.exists(obj)
    syms = subprocess.check_output(["nm", obj]).decode()
    assert " T ProgressiveDP" in syms
    for undefined in ("numberofseqs", "texts", "textsizes", "rotations", "csadp_align_batch"):
        assert (" U %s" % undefined) in syms


def test_no_cpu_fallback_without_device():
    """On a machine without a GPU every compute entry point must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(csa_amd.CsadpError) as e:
        csa_amd.init()
    assert e.value.code == csa_amd.ERR_NO_DEVICE
    with pytest.raises(csa_amd.CsadpError):
        csa_amd.align_batch([([b"ACGT", b"ACGT"], None, None, None)])
    with pytest.raises(csa_amd.CsadpError):
        csa_amd.PairBatch([([b"ACGT", b"ACGT"], None, None, None)])


def test_dropin_adapter_without_device_exits_2_and_computes_nothing(tmp_path):
    """The drop-in adapter's error policy (include/csa_dropin.h): no CPU fallback -- without a device the first gap prints the
    reason and the process exits with 2, in both modes; a program that never reaches a gap (the reference's mode R) is not
    disturbed by the adapter's early start-up thread."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from helpers import ROOT
    exe = str(tmp_path / "dropin_driver")
    subprocess.check_call(["gcc", "-O1", "-fcommon", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "dropin_driver.c"),
                           os.path.join(ROOT, "csa_amd", "csadp_dropin.o"), "-o", exe, "-L", os.path.join(ROOT, "csa_amd"), "-lcsadp",
                           "-Wl,-rpath," + os.path.join(ROOT, "csa_amd"), "-lpthread"])
    for deferred in ("0", "1"):
        out = subprocess.run([exe, deferred, "2", "1", "ACGTACGT", "ACGTTCGT"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert out.returncode == 2
        assert b"no usable gfx950 HIP device" in out.stderr and b"0 0 " not in out.stdout
    # no gap at all (ngaps = 0): nothing to compute, the failed start-up is never mentioned
    out = subprocess.run([exe, "0", "2", "0", "ACGT", "ACGT"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode == 0 and b"drop-in" not in out.stderr


# DeleteGappedColumns runs either as the reference's plain pass or with its candidate scores speculated first (what the
# round driver does over all tasks and threads); CSADP_REFINE_SPECULATE=1 makes the one-task entry points speculate too.
@pytest.fixture(params=["plain", "speculated", "speculated-threads"])
def refine_mode(request, monkeypatch):
    if request.param != "plain":
        monkeypatch.setenv("CSADP_REFINE_SPECULATE", "1" if request.param == "speculated" else "2")
    return request.param


@pytest.mark.parametrize("name", ["tiny_pairs.json", "tiny_families.json"])
def test_host_logic_matches_golden(name, refine_mode):
    """Ordering, seeding, stale-border rule, traceback application, DeleteGappedColumns and the
    path-derived DP score of the product's host code, fills supplied by the oracle."""
    fill = oracle_filler()
    for c in load_golden(name):
        got = csa_amd.debug_align_with_filler(golden_task(c), fill)
        exp = golden_aligned(c)
        assert got["status"] == 0
        assert got["consensus"] == c["consensus"]
        assert got["aligned"] == (exp if exp[0] is not None else None)


def test_host_logic_medium_families_vs_oracle(refine_mode):
    fill = oracle_filler()
    r = rng(123)
    for n, length in [(4, 120), (7, 90), (12, 60), (3, 200), (9, 700)]:
        fam = random_family(r, n, length, mut=0.12, indel=0.08)
        rots = [r.randrange(len(f)) for f in fam]
        got = csa_amd.debug_align_with_filler((fam, rots, None, None), fill)
        cons, strs, st = oracle_progressive(fam, rots)
        assert got["aligned"] == strs and got["consensus"] == cons
        assert got["score"] == st.last_score and got["cells"] == st.cells and got["fills"] == st.fills


def test_round_driver_with_one_two_and_three_groups(monkeypatch):
    """csadp_align_batch's own round driver (lock-step rounds; the tasks dealt longest first over round groups, a host thread
    each) on the CPU, the oracle's fills in place of the device step: the same strings whatever the number of groups, and the
    oracle's.  The switch is read once per process: tests/conftest.py reloads it after every monkeypatched variable."""
    fill = oracle_filler()
    r = rng(321)
    tasks = []
    for n, length in [(4, 150), (3, 40), (6, 90), (5, 1200), (8, 60), (3, 300), (7, 30)]:
        fam = random_family(r, n, length, mut=0.12, indel=0.08)
        tasks.append((fam, [r.randrange(len(f)) for f in fam], None, None))
    want = [oracle_progressive(t[0], t[1]) for t in tasks]
    for groups in ("1", "2", "3"):
        monkeypatch.setenv("CSADP_ROUND_GROUPS", groups)
        got = csa_amd.debug_align_batch_with_filler(tasks, fill)
        for (cons, strs, st), g in zip(want, got):
            assert g["status"] == 0 and g["aligned"] == strs and g["consensus"] == cons and g["fills"] == st.fills, groups


def test_free_results_counts_failures_and_frees_everything():
    """csadp_free_results (one call per batch for streaming callers): frees every result, reports how many carried an
    error; results of the test seam serve as input (one of them fails on its alphabet)."""
    import ctypes
    fill = oracle_filler()
    L = csa_amd.lib()
    res = (csa_amd.Result * 3)()
    tasks = [([b"ACGTACGT", b"ACGAACGT"], None, None, None), ([b"ACGT", b"ACXT"], None, None, None), ([b"AC", b"AC"], None, None, None)]
    cb = csa_amd.DEBUG_FILL_FN(fill)
    for i, t in enumerate(tasks):
        ta = csa_amd.TaskArray([t])
        L.csadp_debug_align_with_filler(ta.arr, cb, None, ctypes.byref(res[i]))
    assert [r.status for r in res] == [0, csa_amd.ERR_ALPHABET, 0]
    assert L.csadp_free_results(res, 3, 2) == 1
    assert all(not r.aligned and not r.progress for r in res)
    assert L.csadp_free_results(res, 3, 2) == 1            # idempotent on freed results
    assert L.csadp_free_results(None, 0, 2) == 0


def test_host_logic_rejects_bad_input():
    fill = oracle_filler()
    assert csa_amd.debug_align_with_filler(([b"ACGT", b"ACXT"], None, None, None), fill)["status"] == csa_amd.ERR_ALPHABET
    assert csa_amd.debug_align_with_filler(([b"ACGT"], None, None, None), fill)["status"] == csa_amd.ERR_ARG
    assert csa_amd.debug_align_with_filler(([b"ACGT", b"ACGT"], [0, 9], None, None), fill)["status"] == csa_amd.ERR_ARG
    # IUPAC letter outside the aligned region is fine (the reference never reads it)
    ok = csa_amd.debug_align_with_filler(([b"ACGTN", b"ACGT"], [0, 0], [0, 0], [4, 4]), fill)
    assert ok["status"] == 0 and ok["aligned"] == [b"ACGT", b"ACGT"]


def test_partition_lpt():
    costs = [9, 7, 6, 5, 4, 3, 2, 2, 1]
    assign, maxload = csa_amd.partition_lpt(costs, 3)
    loads = [sum(c for c, a in zip(costs, assign) if a == p) for p in range(3)]
    assert max(loads) == maxload == 13 and sum(loads) == sum(costs)
    assert csa_amd.partition_lpt([], 4) == ([], 0)
    # config-5 style spread: 256 tasks over 8 GPUs within 5 % of the mean
    la, lb = config5_lengths(256)
    costs = [a * b for a, b in zip(la, lb)]
    assign, maxload = csa_amd.partition_lpt(costs, 8)
    assert maxload <= 1.05 * sum(costs) / 8


def test_fasta_loader_matches_reference_rules(tmp_path):
    for name in ("Primates", "Mammals"):
        path = os.path.join(GOLDEN, "data", name + ".txt")
        got = csa_amd.load_fasta(path)
        _, seqs = read_fasta(path)
        assert [t for _, t in got] == seqs
    p = tmp_path / "odd.fa"
    p.write_bytes(b"junk\n>one desc\nac gt-\r\nNNry\n>bad\nACG*T\n>empty\n\n>two\nTTTT\n")
    got = csa_amd.load_fasta(str(p))
    assert got == [("one desc", b"ACGTNNRY"), ("two", b"TTTT")]
    p.write_bytes(b">only\nACGT\n")
    with pytest.raises(csa_amd.CsadpError):
        csa_amd.load_fasta(str(p))


def test_synthetic_pairs_are_deterministic_and_related():
    a, b, ra, rb = synth_pair(7, length=2000)
    a2, b2, ra2, rb2 = synth_pair(7, length=2000)
    assert (a, b, ra, rb) == (a2, b2, ra2, rb2)
    assert synth_pair(8, length=2000)[0] != a
    cons, strs, st = oracle_progressive([a, b], [ra, rb])
    assert degap(strs[0]) == rotated(a, ra) and degap(strs[1]) == rotated(b, rb)
    assert st.last_score > 0.5 * len(a)          # ~10 % substitutions, 2 % indels


def test_rotated_fasta_wire_format(tmp_path):
    """csadp_write_rotated_fasta reproduces the reference's <base>-Rotated.fasta byte for byte
    (md5 recorded from the unmodified reference program, tests/golden/pipeline.json) and
    csadp_read_rotations recovers the offsets from its headers."""
    import hashlib
    import json
    sys_path = os.path.join(GOLDEN, "make_golden.py")
    assert os.path.exists(sys_path)
    rot = {"Primates": [1947, 1949, 1950, 2530, 1952, 1946, 1951, 1952, 1975, 1955, 1954, 2475, 1948, 1947, 1940, 1948],
           "Mammals": [1283, 1304, 1263, 1640, 1277, 1722, 1295, 1272, 1851, 1273, 1266, 1273]}
    with open(os.path.join(GOLDEN, "pipeline.json")) as f:
        gold = json.load(f)
    for name in ("Primates", "Mammals"):
        recs = csa_amd.load_fasta(os.path.join(GOLDEN, "data", name + ".txt"))
        out = str(tmp_path / (name + "-Rotated.fasta"))
        csa_amd.write_rotated_fasta(out, recs, rot[name])
        with open(out, "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == gold[name]["rotated_md5"]
        assert csa_amd.read_rotations(out) == rot[name]


def test_progress_tokens_equal_the_reference_log(refine_mode):
    """csadp_result.progress restates the reference's stdout tokens between "[(min-max)" and "->":
    one '.' per fill (dynamicprogramming.c:1156) and one '!' per all-gap column DeleteGappedColumns
    meets (:689).  pipeline.json holds the unmodified program's log lines; the Mammals gaps (42 calls,
    two '!') run here through the product's host logic, fills from the oracle (test seam)."""
    import re
    import csa_amd as C
    from helpers import GOLDEN, read_fasta
    gold = load_golden("pipeline.json")["Mammals"]
    _, seqs = read_fasta(os.path.join(GOLDEN, "data", "Mammals.txt"))
    rc, segs, _ = C.build_anchor_map(seqs, gold["rotations"])
    assert rc == 0
    fill = oracle_filler()
    n = len(seqs)
    lines = []
    for k in range(len(segs) - 1):
        size, dp, pos = segs[k]
        if not dp:
            continue
        starts = [p + size for p in pos]
        ends = list(segs[k + 1][2])
        if max(e - s for s, e in zip(starts, ends)) > 1600:
            lines.append(None)                    # the large gaps take the Python walk too long; tokens are per gap
            continue
        got = C.debug_align_with_filler((seqs, gold["rotations"], starts, ends), fill)
        assert got["status"] == 0
        gaps = [e - s for s, e in zip(starts, ends)]
        lines.append("[(%-4d-%4d)%s->%4d]" % (min(gaps), max(gaps), got["progress"], got["consensus"]))
    assert len(lines) == len(gold["dp_log"]) == gold["dp_calls"]
    checked = [(a, b) for a, b in zip(lines, gold["dp_log"]) if a is not None]
    assert len(checked) >= 30
    for a, b in checked:
        assert a == b
    assert sum(a.count("!") for a, _ in checked) >= 1


def test_bench_reads_its_roofline_sources_from_profiles():
    """bench.py's `frac_from_kernel_duration` and `frac_from_counters` must be recomputable from ONE committed file each (round-4 VERDICT,
    item 5): the helper finds the newest round's summaries under profiles/ and the arithmetic is the one DESIGN.md section 5 states."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    row = bench.kernel_stats_row("bench_kernel_solo.csv", "nw_fill_bits<2")
    assert row and row["calls"] >= 10 and 1500 < row["avg_us"] < 3500 and row["file"].startswith("profiles/r0")
    cells = 2 * 128 * 16384 * 16384                       # two passes of the 128-pair batch per launch (roughly: b is a little longer or shorter)
    out = bench.roofline_from_profiles(bench.bits_valu_per_cell(2), cells, 2)
    assert abs(out["frac_from_kernel_duration"] - bench.bits_valu_per_cell(2) * cells / (row["avg_us"] * 1e-6) / 1e12 / bench.VALU_PEAK_TOPS) < 1e-3
    with open(os.path.join(ROOT, out["counters_source"].split(":")[0])) as f:
        pmc = json.load(f)["nw_fill_bits"]
    assert abs(out["frac_from_counters"] - 2.0 * pmc["valu_insts_per_wave"] / pmc["wave_cycles_per_wave_x4"]) < 1e-3
    assert 0.2 < out["frac_from_kernel_duration"] < 0.6 and 0.2 < out["frac_from_counters"] < 0.6
