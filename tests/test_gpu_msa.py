"""GPU tests (-m gpu) of the whole alignment stage behind the C-ABI (csadp_msa, csa_msa): rotation
finder + anchor map on the host, every DP gap in ONE device batch, rows as SaveAlignment prints
them.  Expected bytes come from the compiled reference (tests/golden/anchors.json, pipeline.json)."""
import hashlib
import json
import os
import shutil
import subprocess

import pytest

import csa_amd
import helpers as H

pytestmark = pytest.mark.gpu

TOOL = os.path.join(H.ROOT, "csa_amd", "csa_msa")


def golden():
    with open(os.path.join(H.GOLDEN, "anchors.json")) as f:
        return json.load(f)


def md5(path):
    with open(path, "rb") as f:
        return hashlib.md5(f.read()).hexdigest()


def test_msa_rows_match_reference_fixture():
    """60 small families: rotations (found or given), and every row byte for byte."""
    csa_amd.init(device=0)
    gaps = 0
    for case in golden()["families"]:
        seqs = [s.encode() for s in case["seqs"]]
        rc, rot, rows, st = csa_amd.msa(seqs, case["rotations"] if case["given"] else None)
        assert rc == 0
        assert rot == case["rotations"]
        assert [r.decode() for r in rows] == case["rows"]
        assert st["segments"] == len(case["segments"])
        assert st["dp_gaps"] == sum(s[1] for s in case["segments"])
        gaps += st["dp_gaps"]
    assert gaps > 100


@pytest.mark.parametrize("name", ["Primates", "Mammals", "Set3"])
def test_csa_msa_tool_writes_the_reference_files(name, tmp_path):
    """Mode N on the reference's example sets: -Rotated.fasta and -Aligned.fasta have the md5 of the
    files the unmodified reference program writes, and csadp_sp_score (K3) says about the written
    alignment what the reference's own mode S says (tools.c:194-293; SURVEY 8c's numbers).
    Set3 is the input whose anchoring mostly fails: 36 fills up to 16979 x 20852 with profiles of up
    to 18 sequences (5.4e9 cells) -- the profile-step kernel at full size."""
    want = H.load_golden("pipeline.json")[name]
    src = str(tmp_path / (name + ".txt"))
    shutil.copy(os.path.join(H.GOLDEN, "data", name + ".txt"), src)
    out = subprocess.run([TOOL, "N", src], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert out.returncode == 0, out.stdout.decode()
    log = out.stdout.decode()
    assert md5(str(tmp_path / (name + "-Rotated.fasta"))) == want["rotated_md5"]
    assert md5(str(tmp_path / (name + "-Aligned.fasta"))) == want["aligned_md5"]
    assert "%d gaps by DP" % want["dp_calls"] in log
    with open(str(tmp_path / (name + "-Aligned.fasta")), "rb") as f:
        rows = [ln.rstrip(b"\n") for ln in f if not ln.startswith(b">")]
    st = csa_amd.sp_score(rows)
    assert {"consensus": st["consensus"], "avg_gaps": st["total_gaps"] // len(rows), "conserved": st["conserved_columns"],
            "sp": st["sp_score"]} == want["mode_s"]


def test_msa_mode_a_and_writer(tmp_path):
    """Mode A (rotations all zero) through the library, written with csadp_write_aligned_fasta; the
    rows degap to the inputs and all have one length."""
    csa_amd.init(device=0)
    r = H.rng(77)
    fam = H.random_family(r, 5, 900, mut=0.06, indel=0.02)
    rc, rot, rows, st = csa_amd.msa(fam, [0] * len(fam))
    assert rc == 0 and rot == [0] * len(fam)
    assert len({len(x) for x in rows}) == 1
    for row, seq in zip(rows, fam):
        assert H.degap(row) == seq
    path = str(tmp_path / "x-Aligned.fasta")
    csa_amd.write_aligned_fasta(path, ["s%d" % i for i in range(len(fam))], rot, rows)
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    assert lines[0] == b">s0 @ 0" and lines[1] == rows[0] and lines[-1] == b""


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built")
def test_msa_matches_reference_live(tmp_path):
    csa_amd.init(device=0)
    checked = 0
    for seed in range(12):
        r = H.rng(8100 + seed)
        fam = H.rotated_family(r, r.choice([3, 4, 6]), r.choice([400, 1200]), mut=0.06, indel=0.02)
        path = str(tmp_path / ("r%d.fasta" % seed))
        rc, rot, _, _ = H.ref_alignment_map(fam, savepath=path, timeout=60)
        if rc != 0:
            continue
        with open(path, "rb") as f:
            want = [ln.rstrip(b"\n") for ln in f if not ln.startswith(b">")]
        rc2, rot2, rows, _ = csa_amd.msa(fam)
        assert rc2 == 0 and rot2 == rot
        assert rows == want
        checked += 1
    assert checked >= 6
