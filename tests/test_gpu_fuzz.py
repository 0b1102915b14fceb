"""Differential fuzz (-m gpu): random ProgressiveDP tasks through the HIP path against the COMPILED REFERENCE (oracle/_ref/libcsa_ref.so --
the unmodified dynamicprogramming.c, built in the build container and carried to the GPU box as a binary).  A short run of
tools/r05/fuzz_vs_reference.py per mode; the long runs of the round are recorded in profiles/r05_fuzz_vs_reference.txt."""
import os
import subprocess
import sys

import pytest

from helpers import ROOT, have_ref

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libcsa_ref.so not built (needs /root/reference at build time)")
@pytest.mark.parametrize("mode", ["families", "pairs"])
def test_short_fuzz_against_the_compiled_reference(mode):
    args = [sys.executable, os.path.join(ROOT, "tools", "r05", "fuzz_vs_reference.py"), "12", "99"] + (["pairs"] if mode == "pairs" else [])
    run = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = run.stdout.decode(errors="replace")
    assert run.returncode == 0, out[-3000:]
    last = out.strip().splitlines()[-1]
    assert " 0 mismatches" in last and int(last.split(":")[1].split()[0]) >= 100, last
