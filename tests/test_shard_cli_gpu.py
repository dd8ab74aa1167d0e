"""GPU test of the multi-GPU front end as a user starts it: torch.distributed.run with two ranks (they share the box's
one GPU: EDSX_SHARE_GPU=1, gloo for the object collectives), files in, files out, compared with the oracle."""
import os
import random
import subprocess
import sys

import pytest

import oracle_lib as o
from test_merge_shard_cpu import shaped_eds
from test_vcf_shard_cpu import _free_port, _random_records, _vcf

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(args, nproc=2, backend="gloo"):
    env = dict(os.environ, EDSX_SHARE_GPU="1", EDSX_DIST_BACKEND=backend, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "edsparser_amd.shard"] + args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)


def test_vcf2eds_front_end(tmp_path):
    rng = random.Random(77)
    ref = "".join(rng.choice("ACGT") for _ in range(60000))
    recs = _random_records(rng, ref, 5000, 8)
    vcf, fasta = _vcf(ref, recs, 8)
    (tmp_path / "x.vcf").write_bytes(vcf)
    (tmp_path / "ref.fa").write_bytes(fasta)
    r = _launch(["vcf2eds", "-i", str(tmp_path / "x.vcf"), "-r", str(tmp_path / "ref.fa")])
    assert r.returncode == 0, r.stderr[-2000:]
    want = o.vcf(vcf, fasta, 0)
    assert (tmp_path / "x.eds").read_bytes() == want[0] and (tmp_path / "x.seds").read_bytes() == want[1]
    assert "Variant groups created:     %d" % want[2]["variant_groups"] in r.stdout


def test_eds2leds_front_end_and_error_exit(tmp_path):
    rng = random.Random(78)
    eds, seds = shaped_eds(rng, 2000, 12, True, True)
    (tmp_path / "g.eds").write_bytes(eds)
    (tmp_path / "g.seds").write_bytes(seds)
    r = _launch(["eds2leds", "-i", str(tmp_path / "g.eds"), "-s", str(tmp_path / "g.seds"), "-l", "12"])
    assert r.returncode == 0, r.stderr[-2000:]
    want = o.merge(eds, seds, 12, True)
    assert (tmp_path / "g_l12.leds").read_bytes() == want[0] and (tmp_path / "g_l12.seds").read_bytes() == want[1]
    assert "Symbol ranges: 2" in r.stdout
    (tmp_path / "bad.eds").write_bytes(b"{A,C}{G")
    r = _launch(["eds2leds", "-i", str(tmp_path / "bad.eds"), "-l", "3"])
    assert r.returncode != 0 and "Error: Expected '}'" in r.stderr


@pytest.mark.parametrize("lw", [None, 60])
def test_msa2eds_front_end(tmp_path, lw):
    """Two ranks, each with its column slab of every row of the file; the stitched pieces equal the oracle's output."""
    from msa_cases import random_msa
    rng = random.Random(79)
    msa = random_msa(rng, S=40, L=30000, lw=lw or 10 ** 6, p_var=0.06)
    (tmp_path / "a.msa").write_bytes(msa)
    r = _launch(["msa2eds", "-i", str(tmp_path / "a.msa")])
    assert r.returncode == 0, r.stderr[-2000:]
    want = o.msa(msa, 0)
    assert (tmp_path / "a.eds").read_bytes() == want[0] and (tmp_path / "a.seds").read_bytes() == want[1]
    assert "Column slabs: 2" in r.stdout


def test_msa2eds_front_end_leds(tmp_path):
    """The same with a context length: the boundary between the two slabs is recomputed between their nearest common
    runs of at least l columns (edsx_msa_anchor_info); the pieces equal the oracle's l-EDS."""
    from msa_cases import random_msa
    rng = random.Random(80)
    msa = random_msa(rng, S=40, L=30000, lw=10 ** 6, p_var=0.06)
    (tmp_path / "a.msa").write_bytes(msa)
    r = _launch(["msa2eds", "-i", str(tmp_path / "a.msa"), "-l", "6"])
    assert r.returncode == 0, r.stderr[-2000:]
    want = o.msa(msa, 6)
    assert (tmp_path / "a_l6.leds").read_bytes() == want[0] and (tmp_path / "a_l6.seds").read_bytes() == want[1]
    assert "Column slabs: 2" in r.stdout


def test_rccl_backend_single_rank(tmp_path):
    """The production backend (nccl = RCCL) with the one rank a one-GPU box allows: process-group start-up on the device and
    every tensor collective of the three front ends run through RCCL on device tensors (int64 vectors, uint8 payloads); the
    multi-rank logic itself is what the two-rank gloo tests above and the CPU tests cover."""
    from msa_cases import random_msa
    rng = random.Random(80)
    msa = random_msa(rng, S=30, L=20000, lw=None, p_var=0.06)
    (tmp_path / "a.msa").write_bytes(msa)
    r = _launch(["msa2eds", "-i", str(tmp_path / "a.msa")], nproc=1, backend="nccl")
    assert r.returncode == 0, r.stderr[-2000:]
    want = o.msa(msa, 0)
    assert (tmp_path / "a.eds").read_bytes() == want[0] and (tmp_path / "a.seds").read_bytes() == want[1]
    ref = "".join(rng.choice("ACGT") for _ in range(30000))
    recs = _random_records(rng, ref, 2000, 6)
    vcf, fasta = _vcf(ref, recs, 6)
    (tmp_path / "x.vcf").write_bytes(vcf)
    (tmp_path / "ref.fa").write_bytes(fasta)
    r = _launch(["vcf2eds", "-i", str(tmp_path / "x.vcf"), "-r", str(tmp_path / "ref.fa"), "-l", "5"], nproc=1, backend="nccl")
    assert r.returncode == 0, r.stderr[-2000:]
    want = o.vcf(vcf, fasta, 5)
    assert (tmp_path / "x_l5.leds").read_bytes() == want[0] and (tmp_path / "x_l5.seds").read_bytes() == want[1]
    eds, seds = shaped_eds(rng, 1500, 10, True, True)
    (tmp_path / "g.eds").write_bytes(eds)
    (tmp_path / "g.seds").write_bytes(seds)
    r = _launch(["eds2leds", "-i", str(tmp_path / "g.eds"), "-s", str(tmp_path / "g.seds"), "-l", "10"], nproc=1, backend="nccl")
    assert r.returncode == 0, r.stderr[-2000:]
    want = o.merge(eds, seds, 10, True)
    assert (tmp_path / "g_l10.leds").read_bytes() == want[0] and (tmp_path / "g_l10.seds").read_bytes() == want[1]
