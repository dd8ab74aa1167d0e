"""The C++ multi-GPU path of msa2eds (edsx_msa_transform_multi, csrc/multi_gpu.hip): rank threads inside the library,
column slabs of the host image, boundary stitch.  On the one-GPU box N ranks share the device and exchange in process
(RCCL does not run two ranks on one device); the RCCL exchange itself runs with one rank.  Expected bytes: the oracle."""
import os
import random
import subprocess
import sys

import pytest

import oracle_lib as o
from msa_cases import random_msa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _multi(n, rccl=False):
    import edsparser_amd
    return edsparser_amd.MultiGpu([0] * n, use_rccl=rccl)


@pytest.mark.parametrize("n", [2, 3, 5])
def test_slabs_equal_the_oracle(n):
    m = _multi(n)
    rng = random.Random(100 + n)
    parts = chains = 0
    for i in range(60):
        lw = rng.choice([None, None, 7, 60])
        msa = random_msa(rng, S=rng.randint(2, 9), L=rng.randint(2 * n, 300), lw=lw, trailing_newline=rng.random() < 0.7,
                         p_var=rng.choice([0.02, 0.1, 0.4]))
        got = m.msa_transform(msa, 0)
        assert got == o.msa(msa, 0), (i, msa)
        p, c = m.last_partition()
        parts += p
        chains += c
    assert parts >= 55 and chains > 20                      # the slabs were cut, and runs did cross the cuts


def test_long_runs_make_chains_over_several_slabs():
    m = _multi(4)
    rows = [b"ACGT" * 50, b"ACGT" * 50, b"ACGT" * 50]                          # one common run over all four slabs
    msa = b"".join(b">s%d\n%s\n" % (i, r) for i, r in enumerate(rows))
    assert m.msa_transform(msa, 0) == o.msa(msa, 0)
    rows = [b"A" * 200, b"C" * 200, b"A" * 100 + b"G" * 100]                    # one variant run over all four slabs
    msa = b"".join(b">s%d\n%s\n" % (i, r) for i, r in enumerate(rows))
    assert m.msa_transform(msa, 0) == o.msa(msa, 0)
    assert m.last_partition() == (True, 1)


def test_odd_files_and_short_leds_slabs_are_not_partitioned():
    m = _multi(3)
    rng = random.Random(7)
    msa = random_msa(rng, S=5, L=30)
    for l in (3, 10):                                                         # slabs of ten columns: no room for anchors
        assert m.msa_transform(msa, l) == o.msa(msa, l)
        assert m.last_partition()[0] is False
    import edsparser_amd
    with pytest.raises(edsparser_amd.EdsxError) as ei:
        m.msa_transform(b">a\nACGT\n>b\nACG\n", 0)                            # ragged rows: the transform words the error
    assert "Invalid MSA" in str(ei.value)


@pytest.mark.parametrize("n", [2, 3, 5])
def test_leds_slabs_stitched_between_standalone_runs(n):
    """Context length > 0: every slab boundary is recomputed between the nearest common runs of at least l columns on
    either side (run_rank_leds).  Random alignments with sparse and dense variation, one-line and wrapped rows; cases
    whose slabs have no such runs fall back to one GPU - both kinds must occur and both equal the oracle."""
    m = _multi(n)
    rng = random.Random(300 + n)
    parts = wholes = 0
    for i in range(80):
        lw = rng.choice([None, None, 7, 60])
        l = rng.choice([1, 2, 3, 5, 8, 16])
        msa = random_msa(rng, S=rng.randint(2, 9), L=rng.randint(40 * n, 900), lw=lw, trailing_newline=rng.random() < 0.7,
                         p_var=rng.choice([0.01, 0.03, 0.1, 0.3]))
        got = m.msa_transform(msa, l)
        assert got == o.msa(msa, l), (i, l, msa)
        p, c = m.last_partition()
        parts += p
        wholes += not p
        if p:
            assert c == n - 1
    assert parts >= 30 and wholes >= 3


def test_leds_anchor_edge_cases():
    """Anchors that touch the slab edges: a common run that spans a boundary (standalone as a whole, maybe not by its
    pieces), slabs that begin or end with their anchor, variant columns right at the cut.  Lower-case letters of the
    template are the variant columns (the second row has another base there)."""
    m = _multi(2)
    A = "ACGTACGTACGTACGTACGT"                                      # 20 common columns
    other = {"a": "C", "c": "G", "g": "T", "t": "A"}
    templates = [A + "t" + A + A + "g" + A,                        # the cut falls inside a long common run
                 A + "t" + A[:3] + A[:2] + "g" + A + "A",          # ... inside a short one
                 A + "tt" + "gg" + A,                              # variant columns on both sides of the cut
                 A + "t" + A + "g" + A + "c" + A + "AAA",
                 "t" + A + A + "g",                                # variant columns at the ends of the alignment
                 A + A + "c" + A + A]                              # one variant column exactly at the cut (len 81 -> cut at 40)
    partitioned = 0
    for l in (1, 4, 10, 20, 21):
        for t in templates:
            r0 = t.upper().encode()
            r1 = "".join(other[ch] if ch.islower() else ch for ch in t).encode()
            msa = b"".join(b">s%d\n%s\n" % (i, r) for i, r in enumerate([r0, r1, r0]))
            assert m.msa_transform(msa, l) == o.msa(msa, l), (l, t)
            partitioned += m.last_partition()[0]
    assert partitioned >= 10


def test_larger_alignment_device_generated():
    """64 x 200 k columns generated on the device, five slabs against the single call."""
    import torch
    import edsparser_amd
    ctx = edsparser_amd.Context(0)
    S, L = 64, 200_000
    n = edsparser_amd.synth_size(S, L)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    ctx.msa_synth_device(buf.data_ptr(), n, S, L)
    torch.cuda.synchronize()
    host = bytes(buf.cpu().numpy())
    want = ctx.msa_transform(host, 0)
    m = _multi(5)
    assert m.msa_transform(host, 0) == want
    assert m.last_partition()[0] is True


def test_rccl_exchange_single_rank():
    """ncclCommInitAll / ncclAllGather from C++ with the one rank a one-GPU box allows (world 1 is not partitioned, so the
    call goes through the exchange object only at creation); two distinct devices are needed for more."""
    m = _multi(1, rccl=True)
    msa = random_msa(random.Random(3), S=4, L=100)
    assert m.msa_transform(msa, 0) == o.msa(msa, 0)
    import edsparser_amd
    with pytest.raises(edsparser_amd.EdsxError):
        edsparser_amd.MultiGpu([0, 0], use_rccl=True)                          # RCCL: one distinct device per rank


def test_msa2eds_cli_gpus_option(tmp_path):
    from test_host_cpp import BUILD, _build_host
    _build_host()
    msa = random_msa(random.Random(11), S=6, L=500)
    (tmp_path / "x.msa").write_bytes(msa)
    r = subprocess.run([os.path.join(BUILD, "msa2eds"), "-i", str(tmp_path / "x.msa"), "--gpus", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    e, s = o.msa(msa, 0)
    assert (tmp_path / "x.eds").read_bytes() == e and (tmp_path / "x.seds").read_bytes() == s
    assert "GPUs: 1" in r.stdout
    r = subprocess.run([os.path.join(BUILD, "msa2eds"), "-i", str(tmp_path / "x.msa"), "--gpus", "99"], capture_output=True, text=True)
    assert r.returncode == 1 and "Error:" in r.stderr


def test_msa2eds_cli_batches_option(tmp_path):
    from test_host_cpp import BUILD, _build_host
    _build_host()
    msa = random_msa(random.Random(12), S=6, L=900, lw=60)
    (tmp_path / "x.msa").write_bytes(msa)
    for l in (0, 4):
        args = [os.path.join(BUILD, "msa2eds"), "-i", str(tmp_path / "x.msa"), "--batches", "3"] + (["-l", str(l)] if l else [])
        r = subprocess.run(args, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        e, s = o.msa(msa, l)
        base = "x_l%d" % l if l else "x"
        assert (tmp_path / (base + (".leds" if l else ".eds"))).read_bytes() == e
        assert (tmp_path / (base + ".seds")).read_bytes() == s
        assert "Column batches: " in r.stdout
    r = subprocess.run([os.path.join(BUILD, "msa2eds"), "-i", str(tmp_path / "x.msa"), "--batches", "0"], capture_output=True, text=True)
    assert r.returncode == 1 and "Error:" in r.stderr


def test_anchor_info_equals_the_cpu_restatement():
    """edsx_msa_anchor_info (the first / last common segment of at least l columns of a planned slab, with its text
    offsets) against the same numbers derived on the CPU from the slab's rows and the oracle's text."""
    import edsparser_amd
    from test_multigpu_cpu import anchors_of, rows_of
    ctx = edsparser_amd.Context(0)
    rng = random.Random(41)
    found = 0
    for i in range(120):
        l = rng.choice([1, 2, 4, 9, 30])
        msa = random_msa(rng, S=rng.randint(2, 8), L=rng.randint(5, 500), lw=rng.choice([None, 7, 60]), p_var=rng.choice([0.02, 0.1, 0.4]))
        eds, seds = ctx.msa_transform(msa, l)
        assert (eds, seds) == o.msa(msa, l)
        got = ctx.msa_anchor_info(l)
        want = anchors_of(rows_of(msa), l, eds, seds)
        if not want["found"]:
            assert not got["found"], (i, l)
            continue
        found += 1
        assert got == want, (i, l, got, want)
    assert found > 60
    ctx.close()
