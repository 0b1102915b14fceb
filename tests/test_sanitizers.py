"""The product's HOST code under sanitizers, CPU only: tests/sanitize/host_sanitize.cpp is built twice
from the product's own host sources -- with -fsanitize=address,undefined and with -fsanitize=thread -- and drives
the FASTA reader/writers, the rotation finder, the anchor map, the whole progressive host logic (test seam, fills
from the oracle; results compared with the oracle) and the host thread pool from several threads.  A sanitizer
report makes the binary exit non-zero."""
import os
import subprocess

import pytest

from helpers import GOLDEN, ROOT


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_host_code_under_sanitizers(kind, tmp_path):
    if not os.path.exists(os.path.join(ROOT, "build", "obj", "csadp_cells.o")):
        pytest.skip("device objects not built (run __graft_entry__.build())")
    make = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "sanitize"), kind], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, timeout=900)
    assert make.returncode == 0, make.stdout.decode()[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    # twice: DeleteGappedColumns as the plain pass, and with its candidate scores speculated on several threads first
    for refine in ("0", "2"):
        run = subprocess.run([os.path.join(ROOT, "build", "host_" + kind), os.path.join(GOLDEN, "data", "Primates.txt"), str(tmp_path)],
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=dict(env, CSADP_REFINE_SPECULATE=refine))
        out = run.stdout.decode(errors="replace")
        assert run.returncode == 0, out[-4000:]
        assert "host_sanitize ok" in out
        assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out and "WARNING: ThreadSanitizer" not in out
