"""Drop-in test (-m gpu): the reference PROGRAM with its DP translation unit replaced by
csa_amd/csrc/csadp_dropin.c + libcsadp.so (binary oracle/_ref/CSA_csadp, built in the build
container by `make -C oracle _dropin` from the reference sources where they lie) must write
the same <set>-Aligned.fasta as the unmodified reference (tests/golden/pipeline.json).
This drives ProgressiveDP exactly as RunAlignment does (alignment.c:201): ~50 calls per set,
up to 16 sequences each, stale borders, DeleteGappedColumns, empty regions."""
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


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/CSA_csadp not built (needs /root/reference at build time)")
@pytest.mark.parametrize("name", ["Primates", "Mammals", "Set3"])
def test_reference_program_with_csadp_dropin(name, tmp_path):
    with open(os.path.join(GOLDEN, "pipeline.json")) as f:
        gold = json.load(f)[name]
    shutil.copy(os.path.join(GOLDEN, "data", name + ".txt"), tmp_path)
    with open(os.devnull) as devnull:
        run = subprocess.run([BIN, name + ".txt"], cwd=tmp_path, stdin=devnull, stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, timeout=600)
    log = run.stdout.decode(errors="replace")
    assert run.returncode == 0, log[-2000:]
    assert log.count("[(") == gold["dp_calls"]
    # the stdout tokens of every call -- "[(min-max)", one '.' per fill (:1156), '!' per all-gap column
    # DeleteGappedColumns met (:689), "->consensus]" -- equal the unmodified program's, so logs diff cleanly
    assert re.findall(r"\[\([^\]]*\]", log) == gold["dp_log"]
    with open(tmp_path / (name + "-Aligned.fasta"), "rb") as f:
        assert hashlib.md5(f.read()).hexdigest() == gold["aligned_md5"], log[-2000:]
    # the reference's own integrity check (tools.c:123-191) ran on our strings and said OK
    assert "Checking integrity of aligned sequences... OK" in log
