"""GPU tests (-m gpu) of the pieces around the DP: the C harness csa_pairs (host in C over the
C-ABI, configs 1-3 of SURVEY.md 8d) and the column-statistics kernel (tools.c:194-293)."""
import os
import re
import subprocess

import pytest

import csa_amd
from helpers import GOLDEN, ROOT, load_golden, random_family, rng, sp_score, oracle_progressive

pytestmark = pytest.mark.gpu

HARNESS = os.path.join(ROOT, "csa_amd", "csa_pairs")
ROT = {"Primates": "1947,1949,1950,2530,1952,1946,1951,1952,1975,1955,1954,2475,1948,1947,1940,1948",
       "Mammals": "1283,1304,1263,1640,1277,1722,1295,1272,1851,1273,1266,1273"}


def _run(args):
    out = subprocess.run([HARNESS] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert out.returncode == 0, out.stdout.decode()
    return out.stdout.decode()


def test_csa_pairs_config1_single_pair():
    """Config 1/2: Primates pair (0,1) with fixture rotations -> len 16589 / SP 15197 / 7ee50a99."""
    log = _run([os.path.join(GOLDEN, "data", "Primates.txt"), "--rot", ROT["Primates"], "--pair", "0,1"])
    assert "pair 0 1 len 16589 SP 15197 score 15197 fnv1a 7ee50a99" in log


def test_csa_pairs_config3_mammals_all_vs_all(tmp_path):
    """Config 3: all 66 Mammals pairs, rotations read back from a -Rotated.fasta the harness wrote."""
    fasta = os.path.join(GOLDEN, "data", "Mammals.txt")
    rotated = str(tmp_path / "Mammals-Rotated.fasta")
    _run([fasta, "--rot", ROT["Mammals"], "--pair", "0,1", "--write-rotated", rotated])
    log = _run([fasta, "--rotated", rotated])
    gold = {(c["a"], c["b"]): c for c in load_golden("real_pairs.json") if c["set"] == "Mammals"}
    seen = 0
    for m in re.finditer(r"pair (\d+) (\d+) len (\d+) SP (-?\d+) score (-?\d+) fnv1a ([0-9a-f]{8})", log):
        a, b = int(m.group(1)), int(m.group(2))
        g = gold[(a, b)]
        assert (int(m.group(3)), int(m.group(4)), int(m.group(5)), m.group(6)) == (g["consensus"], g["sp"], g["sp"], g["fnv1a"])
        seen += 1
    assert seen == 66
    assert "18593884141 cells" in log


def test_sp_score_kernel_matches_tools_rule():
    csa_amd.init(device=0)
    r = rng(5)
    # aligned strings from real progressive tasks (gaps, conserved columns) ...
    cases = [c for c in load_golden("tiny_families.json") if c["aligned"][0]][:40]
    for c in cases:
        strs = [a.encode() for a in c["aligned"]]
        st = csa_amd.sp_score(strs)
        assert st["sp_score"] == sp_score(strs)
        assert st["consensus"] == len(strs[0])
        assert st["total_gaps"] == sum(s.count(b"-") for s in strs)
        assert st["conserved_columns"] == sum(1 for col in zip(*strs) if len(set(col)) == 1)
    # ... and a long random one with IUPAC letters (compared by character, tools.c:276-278)
    fam = [bytes(r.choice(b"ACGT-N") for _ in range(30000)) for _ in range(12)]
    st = csa_amd.sp_score(fam)
    assert st["sp_score"] == sp_score(fam)
    with pytest.raises(csa_amd.CsadpError):
        csa_amd.sp_score([b"ACGT", b"ACG"])


def _mode_s(st, nseq):
    """csadp_sp_stats in the shape of the goldens' "mode_s" records (the tool prints total gaps / nseq, tools.c:284)."""
    return {"consensus": st["consensus"], "avg_gaps": st["total_gaps"] // nseq, "conserved": st["conserved_columns"],
            "sp": st["sp_score"]}


def test_sp_score_kernel_equals_reference_mode_s():
    """f-4 pinned to the reference PROGRAM: sp_stats.json holds the four numbers `CSA S` printed for
    100 small alignments (rows of real runs + adversarial columns: all-gap, IUPAC letters, 64 rows)."""
    csa_amd.init(device=0)
    for case in load_golden("sp_stats.json"):
        rows = [r.encode() for r in case["rows"]]
        assert _mode_s(csa_amd.sp_score(rows), len(rows)) == case["mode_s"]


def test_score_pairs_matches_alignment_scores():
    """csadp_score_pairs == score of csadp_align_batch == SP of the strings, incl. empty regions."""
    csa_amd.init(device=0)
    cases = load_golden("tiny_pairs.json")
    tasks = [([t.encode() for t in c["texts"]], c["rots"], c["starts"], c["ends"]) for c in cases]
    scores, status = csa_amd.score_pairs(tasks)
    full = csa_amd.align_batch(tasks)
    assert status == [0] * len(tasks)
    assert scores == [g["score"] for g in full]
    from csa_amd.synth import synth_pair
    big = [synth_pair(p, length=4000) for p in range(5)]
    tasks = [([a, b], [ra, rb], None, None) for a, b, ra, rb in big]
    scores, status = csa_amd.score_pairs(tasks)
    assert scores == [sp_score(g["aligned"]) for g in csa_amd.align_batch(tasks)]


def test_csa_pairs_native_rotations_end_to_end(tmp_path):
    """Config 1 without any fixture: FASTA -> native rotation finder -> GPU DP; the harness also
    writes the -Rotated.fasta the reference's mode R would write (md5 in pipeline.json)."""
    import hashlib
    import json
    out = str(tmp_path / "Primates-Rotated.fasta")
    log = _run([os.path.join(GOLDEN, "data", "Primates.txt"), "--find-rotations", "--pair", "0,1", "--write-rotated", out])
    assert "rotations: 1947 1949 1950 2530" in log
    assert "pair 0 1 len 16589 SP 15197 score 15197 fnv1a 7ee50a99" in log
    with open(os.path.join(GOLDEN, "pipeline.json")) as f:
        gold = json.load(f)["Primates"]
    with open(out, "rb") as f:
        assert hashlib.md5(f.read()).hexdigest() == gold["rotated_md5"]


def test_score_pairs_device_scores_on_hard_inputs():
    """Score-only calls take the path score from the traceback kernel (checkpoint mode): check it
    on inputs whose paths hug borders, tie everywhere or wander far from the diagonal, on wide jobs
    and on the benchmark pairs, against the score of the full alignment."""
    csa_amd.init(device=0)
    r = rng(31)
    core = bytes(r.choice(b"ACGT") for _ in range(3000))
    junk = bytes(r.choice(b"AC") for _ in range(2500))
    wide = bytes(r.choice(b"ACGT") for _ in range(40000))
    tasks = [
        ([core, junk + core], None, None, None),
        ([core[:1500] + junk + core[1500:], core], None, None, None),
        ([b"G" * 2100, b"T" * 2300], None, None, None),
        ([b"GT" * 1100, b"TG" * 1200], None, None, None),
        ([b"A" * 5000, b"A" * 4993], None, None, None),
        ([b"G", core], None, None, None),
        ([core, core], [17, 1234], None, None),
        ([wide[:700], wide], None, None, None),
        ([wide[100:36000], wide], [5, 9], None, None),
    ]
    from csa_amd.synth import synth_pair
    for p in range(6):
        a, b, ra, rb = synth_pair(p)
        tasks.append(([a, b], [ra, rb], None, None))
    scores, status = csa_amd.score_pairs(tasks)
    full = csa_amd.align_batch(tasks)
    assert status == [0] * len(tasks)
    assert scores == [g["score"] for g in full]
    assert scores[-6:] == [c["sp"] for c in load_golden("config4_pairs.json")[:6]]


# ---- more than one device from one process (csadp_align_batch_on / _multi, csa_pairs --gpus) ------------------

def test_device_ordinals_and_explicit_device():
    """csadp_device_count; an ordinal that does not exist is CSADP_ERR_NO_DEVICE (never silently another GPU);
    the batch on an explicitly named device equals the batch on the primary one; with more than one GPU visible
    the LAST device gives the same bytes as device 0 (every entry point re-binds its device on the calling
    thread -- hipSetDevice is per host thread)."""
    csa_amd.init(device=0)
    n = csa_amd.device_count()
    assert n >= 1
    r = rng(31)
    tasks = []
    for k, length in [(2, 700), (4, 300), (2, 64), (7, 150)]:
        fam = random_family(r, k, length, mut=0.1, indel=0.05)
        fam = [f if f else b"A" for f in fam]
        tasks.append((fam, [r.randrange(len(f)) for f in fam], None, None))
    base = csa_amd.align_batch(tasks)
    assert all(b["status"] == 0 for b in base)
    with pytest.raises(csa_amd.CsadpError) as e:
        csa_amd.align_batch_on(n, tasks)
    assert e.value.code == csa_amd.ERR_NO_DEVICE
    for dev in sorted({0, n - 1}):
        got = csa_amd.align_batch_on(dev, tasks)
        assert [g["aligned"] for g in got] == [b["aligned"] for b in base]
        assert [g["score"] for g in got] == [b["score"] for b in base]


def test_align_batch_multi_equals_single_device():
    """csadp_align_batch_multi: LPT partition + one host thread per device + results in task order.  On a one-GPU
    box both threads drive device 0 (two batches on one engine at once: pools, streams, the host thread pool);
    with several GPUs visible every device takes a part.  Results must equal the single-call results."""
    csa_amd.init(device=0)
    n = csa_amd.device_count()
    r = rng(77)
    tasks = []
    for i in range(24):
        k = r.choice([2, 2, 2, 3, 5])
        fam = random_family(r, k, r.choice([40, 300, 900, 2500]), mut=0.1, indel=0.04)
        fam = [f if f else b"C" for f in fam]
        tasks.append((fam, [r.randrange(len(f)) for f in fam], None, None))
    base = csa_amd.align_batch(tasks)
    devices = list(range(n)) if n > 1 else [0, 0]
    got, st = csa_amd.align_batch_multi(tasks, devices)
    assert [g["aligned"] for g in got] == [b["aligned"] for b in base]
    assert [g["score"] for g in got] == [b["score"] for b in base]
    assert st["ndevices"] == len(devices) and sum(st["tasks"]) == len(tasks)
    assert sum(st["cost"]) == st["total_cost"] == sum(csa_amd.task_cost(t) for t in tasks)
    assert st["max_cost"] == max(st["cost"]) and min(st["tasks"]) >= 1


def test_align_batch_multi_with_eight_logical_devices(monkeypatch):
    """The one-process form of the 8-GPU run (`csa_pairs --gpus 8`, csadp_align_batch_multi with 8 devices), which no round could
    measure on an 8-GPU node: on a box with fewer GPUs CSADP_SHARE_DEVICE=1 maps the ordinals onto the visible ones, so eight host
    threads each take their LPT share of a mixed batch (pairs of 300 .. 20 000 letters, families of 3 .. 6 sequences) through
    csadp_align_batch_on.  Results in task order, equal to the single call; the split within LPT's bound; every device has work."""
    csa_amd.init(device=0)
    if csa_amd.device_count() < 8:
        monkeypatch.setenv("CSADP_SHARE_DEVICE", "1")
    r = rng(808)
    tasks = []
    for i in range(96):
        if i % 4 == 3:
            fam = random_family(r, r.choice([3, 4, 6]), r.choice([200, 1200, 3000]), mut=0.1, indel=0.04)
        else:
            fam = random_family(r, 2, r.choice([300, 2000, 6000, 20000]), mut=0.1, indel=0.03)
        fam = [f if f else b"G" for f in fam]
        tasks.append((fam, [r.randrange(len(f)) for f in fam], None, None))
    base = csa_amd.align_batch(tasks)
    got, st = csa_amd.align_batch_multi(tasks, list(range(8)))
    assert [g["status"] for g in got] == [0] * len(tasks)
    assert [g["aligned"] for g in got] == [b["aligned"] for b in base]
    assert [g["score"] for g in got] == [b["score"] for b in base]
    assert st["ndevices"] == 8 and sum(st["tasks"]) == len(tasks) and min(st["tasks"]) >= 1
    # a few 20 000-letter pairs carry most of the cost: LPT's own bound (mean + the largest task), not 5 %
    assert st["max_cost"] <= st["total_cost"] / 8 + max(csa_amd.task_cost(t) for t in tasks)


def test_csa_pairs_over_two_devices(monkeypatch):
    """The C harness with --gpus 2: csadp_align_batch_multi from plain C.  CSADP_SHARE_DEVICE=1 lets the second
    ordinal fall back onto the one GPU of this box (rehearsal switch); the 66 Mammals pairs must still equal the
    reference digests and the LPT split must be balanced."""
    if csa_amd.device_count() < 2:
        monkeypatch.setenv("CSADP_SHARE_DEVICE", "1")
    log = _run([os.path.join(GOLDEN, "data", "Mammals.txt"), "--rot", ROT["Mammals"], "--gpus", "2"])
    gold = {(c["a"], c["b"]): c for c in load_golden("real_pairs.json") if c["set"] == "Mammals"}
    seen = 0
    for m in re.finditer(r"pair (\d+) (\d+) len (\d+) SP (-?\d+) score (-?\d+) fnv1a ([0-9a-f]{8})", log):
        g = gold[(int(m.group(1)), int(m.group(2)))]
        assert (int(m.group(3)), int(m.group(4)), int(m.group(5)), m.group(6)) == (g["consensus"], g["sp"], g["sp"], g["fnv1a"])
        seen += 1
    assert seen == 66
    m = re.search(r"over 2 GPUs: imbalance ([0-9.]+)", log)
    assert m and float(m.group(1)) <= 1.05
    assert log.count("> gpu ") == 2


def test_bench_collectives_run_over_rccl_on_one_rank():
    """bench.py's multi-GPU plumbing with the REAL backend: CSADP_DIST_FORCE_GROUP=1 makes the single rank
    create an nccl (= RCCL) process group, so the LPT broadcast, the barriers, the max/sum reductions and the
    all-gather of the 16-byte records run on device tensors; the gathered records must cover the whole list."""
    import json
    import sys
    env = dict(os.environ, CSADP_DIST_FORCE_GROUP="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--mode", "strong", "--workload", "config4", "--len", "1024",
                        "--no-cpu-baseline", "--no-extra-legs"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["verified"] is True
    assert line["records"]["gathered"] == 1024
    assert line["records"]["rows_on_rank0"] == 1024 and line["records"]["rows_match_their_records"] is True
    assert line["config"]["lpt_imbalance"] == 1.0


def test_bench_shards_a_real_fasta_batch_over_rccl():
    """--workload mammals (BASELINE config 3): rank 0 reads Mammals.txt and the reference's rotations, broadcasts them as
    one packed pool (RCCL: the forced one-rank group), every rank builds its LPT share of the 66 pairs from the pool,
    the aligned rows are gathered to rank 0 and all 66 records equal the compiled reference's digests."""
    import json
    import sys
    env = dict(os.environ, CSADP_DIST_FORCE_GROUP="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
                        "--mode", "strong", "--workload", "mammals", "--no-cpu-baseline", "--no-extra-legs"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["verified"] is True
    assert line["records"]["gathered"] == 66 and line["records"]["checked_against_reference_digests"] == 66
    assert line["records"]["rows_on_rank0"] == 66 and line["records"]["rows_match_their_records"] is True


def test_warmup_pays_the_first_batch_costs_and_changes_no_result():
    """csadp_warmup (include/csadp.h): the code objects, the copy paths, the host pool and the arenas of csadp_align_batch are paid for up
    front by a small synthetic batch through every kernel family.  It may be called any number of times, before or between batches, and
    the batches around it return the oracle's strings; csadp_last_batch_phases / csadp_recoveries report on the last real batch."""
    csa_amd.init(device=0)
    r = rng(55)
    fam = random_family(r, 5, 800, mut=0.1, indel=0.04)
    tasks = [(fam, [r.randrange(len(f)) for f in fam], None, None), (random_family(r, 2, 3000), None, None, None)]
    want = [oracle_progressive(t[0], t[1]) for t in tasks]
    for _ in range(2):
        csa_amd.warmup()
        got = csa_amd.align_batch(tasks[:1]) + csa_amd.align_batch(tasks[1:])
        for g, (cons, strs, st) in zip(got, want):
            assert g["status"] == 0 and g["aligned"] == strs and g["score"] == st.last_score
    before = csa_amd.recoveries()                          # process-wide since init: other tests force recoveries on purpose
    got = csa_amd.align_batch([tasks[0], tasks[0]])
    ph = csa_amd.last_batch_phases()
    assert ph["tasks"] == 2 and ph["rounds"] == 8 and ph["round_groups"] == 2 and ph["device_ms"] > 0 and ph["wall_ms"] >= ph["seed_ms"]
    assert csa_amd.recoveries() == before
