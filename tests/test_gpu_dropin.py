"""Drop-in test (-m gpu): the reference PROGRAM with its DP translation unit replaced by
csa_amd/csrc/csadp_dropin.c + libcsadp.so (binary oracle/_ref/CSA_csadp, built in the build
container by `make -C oracle _dropin` from the reference sources where they lie) must write
the same <set>-Aligned.fasta as the unmodified reference (tests/golden/pipeline.json).
This drives ProgressiveDP exactly as RunAlignment does (alignment.c:201): ~50 calls per set,
up to 16 sequences each, stale borders, DeleteGappedColumns, empty regions.

Both modes of the adapter (include/csa_dropin.h): CSA_csadp = synchronous, one one-task batch per call;
CSA_csadp_deferred = the SAME sources with one more link flag (-Wl,--wrap=SaveAlignment): the calls record
their gaps and one csadp_align_batch in front of SaveAlignment computes them all."""
import hashlib
import json
import os
import re
import shutil
import subprocess

import pytest

from helpers import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "oracle", "_ref", "CSA_csadp")
BINS = {"synchronous": BIN, "deferred": BIN + "_deferred"}
_logs = {}


def run_program(binary, name, tmp_path, env=None):
    shutil.copy(os.path.join(GOLDEN, "data", name + ".txt"), tmp_path)
    e = dict(os.environ)
    e["CSADP_DROPIN_STATS"] = str(tmp_path / "stats.json")
    e.update(env or {})
    with open(os.devnull) as devnull:
        run = subprocess.run([binary, name + ".txt"], cwd=tmp_path, stdin=devnull, stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, timeout=600, env=e)
    with open(tmp_path / "stats.json") as f:
        stats = json.loads(f.read().splitlines()[-1])
    return run, run.stdout.decode(errors="replace"), stats


@pytest.mark.skipif(not all(os.path.exists(b) for b in BINS.values()),
                    reason="oracle/_ref/CSA_csadp[_deferred] not built (needs /root/reference at build time)")
@pytest.mark.parametrize("mode", ["synchronous", "deferred"])
@pytest.mark.parametrize("name", ["Primates", "Mammals", "Set3"])
def test_reference_program_with_csadp_dropin(name, mode, tmp_path):
    with open(os.path.join(GOLDEN, "pipeline.json")) as f:
        gold = json.load(f)[name]
    run, log, stats = run_program(BINS[mode], name, tmp_path)
    assert run.returncode == 0, log[-2000:]
    # the adapter's own account: which mode ran, one batch per call or one batch for all calls
    assert stats["mode"] == mode and stats["calls"] == gold["dp_calls"]
    assert stats["batches"] == (gold["dp_calls"] if mode == "synchronous" else 1)
    # the WHOLE stdout of the two modes is the same text (RunAlignment prints nothing between two gaps, and the deferred lines
    # come out in call order in front of SaveAlignment's own line)
    _logs[(name, mode)] = log
    other = _logs.get((name, "deferred" if mode == "synchronous" else "synchronous"))
    if other is not None:
        assert other == log
    assert log.count("[(") == gold["dp_calls"]
    # the stdout tokens of every call -- "[(min-max)", one '.' per fill (:1156), '!' per all-gap column
    # DeleteGappedColumns met (:689), "->consensus]" -- equal the unmodified program's, so logs diff cleanly
    assert re.findall(r"\[\([^\]]*\]", log) == gold["dp_log"]
    with open(tmp_path / (name + "-Aligned.fasta"), "rb") as f:
        assert hashlib.md5(f.read()).hexdigest() == gold["aligned_md5"], log[-2000:]
    # the reference's own integrity check (tools.c:123-191) ran on our strings and said OK
    assert "Checking integrity of aligned sequences... OK" in log


@pytest.mark.skipif(not os.path.exists(BINS["deferred"]), reason="oracle/_ref/CSA_csadp_deferred not built")
def test_deferred_binary_can_be_kept_synchronous(tmp_path):
    """CSADP_DROPIN_DEFER=0 keeps a binary linked with --wrap=SaveAlignment in the synchronous mode (include/csa_dropin.h)."""
    with open(os.path.join(GOLDEN, "pipeline.json")) as f:
        gold = json.load(f)["Mammals"]
    run, log, stats = run_program(BINS["deferred"], "Mammals", tmp_path, env={"CSADP_DROPIN_DEFER": "0"})
    assert run.returncode == 0, log[-2000:]
    assert stats["mode"] == "synchronous" and stats["batches"] == gold["dp_calls"]
    assert re.findall(r"\[\([^\]]*\]", log) == gold["dp_log"]
    with open(tmp_path / "Mammals-Aligned.fasta", "rb") as f:
        assert hashlib.md5(f.read()).hexdigest() == gold["aligned_md5"]


def test_explicit_deferred_mode_of_the_adapter(tmp_path):
    """include/csa_dropin.h: csadp_dropin_defer(1) ... csadp_dropin_finish() -- the source-line form of the deferred mode.  A
    small C program (tests/dropin_driver.c) plays RunAlignment: six gaps over five sequences through csadp_dropin.o, once
    synchronous, once deferred; the strings of both are the oracle's."""
    from helpers import oracle_progressive, random_family, rng
    exe = str(tmp_path / "dropin_driver")
    subprocess.check_call(["gcc", "-O1", "-fcommon", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "dropin_driver.c"),
                           os.path.join(ROOT, "csa_amd", "csadp_dropin.o"), "-o", exe, "-L", os.path.join(ROOT, "csa_amd"), "-lcsadp",
                           "-Wl,-rpath," + os.path.join(ROOT, "csa_amd"), "-lpthread"])
    r = rng(31)
    fam = random_family(r, 5, 900, mut=0.1, indel=0.04)
    ngaps = 6
    rots = [(7 * s) % len(f) for s, f in enumerate(fam)]
    want = []
    for g in range(ngaps):
        starts = [len(f) * g // ngaps for f in fam]
        ends = [len(f) * (g + 1) // ngaps for f in fam]
        cons, strs, _ = oracle_progressive(fam, rots, starts, ends)
        want += ["%d %d %s" % (g, s, strs[s].decode()) for s in range(len(fam))]
    for deferred in (0, 1):
        out = subprocess.run([exe, str(deferred), str(len(fam)), str(ngaps)] + [f.decode() for f in fam], stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, timeout=300)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        lines = [ln for ln in out.stdout.decode().splitlines() if not ln.startswith("[(")]
        log = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("[(")]
        assert lines == want
        assert len(log) == ngaps and all(ln.count(".") == len(fam) - 1 for ln in log)
