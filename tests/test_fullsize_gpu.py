"""BASELINE configs[2] and configs[3] at FULL size inside the GPU suite (inputs generated on the device, outputs compared
byte for byte with the CPU oracle), so that the > 2^32 offsets and GB-sized tables of the merge and VCF paths are pinned
by the suite and not by builder logs; plus the device VCF + FASTA generator (csrc/genvcf.hip) at small sizes.
The two full-size tests need ~20 GB of free HBM and about a minute of oracle time on one host core each."""
import os
import sys

import pytest

import oracle_lib as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import edsparser_amd
    c = edsparser_amd.Context(0)
    yield c
    c.close()


def _free_hbm_gb():
    import torch
    free, _total = torch.cuda.mem_get_info(0)
    return free / 2**30


def _records(vcf):
    return [ln.split(b"\t") for ln in vcf.split(b"\n") if ln and not ln.startswith(b"#")]


def test_genvcf_shape_and_parity(ctx):
    vcf, fasta = ctx.genvcf(60_000, 3000, 8, seed=42)
    assert (vcf, fasta) == ctx.genvcf(60_000, 3000, 8, seed=42) and vcf != ctx.genvcf(60_000, 3000, 8, seed=43)[0]
    lines = fasta.split(b"\n")
    assert lines[0] == b">chr1 synthetic" and lines[-1] == b"" and all(len(x) == 60 for x in lines[1:-2]) and 0 < len(lines[-2]) <= 60
    ref = b"".join(lines[1:])
    assert len(ref) == 60_000 and set(ref) <= set(b"ACGT")
    recs = _records(vcf)
    pos = [int(r[1]) for r in recs]
    assert len(recs) == 3000 and all(a < b for a, b in zip(pos, pos[1:])) and all(len(r) == 9 + 8 for r in recs)
    kinds = {"snp": 0, "ins": 0, "del": 0}
    for r in recs:
        p, rf, alt = int(r[1]), r[3], r[4]
        assert ref[p - 1:p - 1 + len(rf)] == rf                      # REF is the reference text at POS
        kinds["snp" if len(rf) == len(alt) == 1 else "ins" if len(alt) > 1 else "del"] += 1
        assert all(g in (b"0|0", b"0|1", b"1|0", b"1|1") for g in r[9:])
    assert 0.6 < kinds["snp"] / 3000 < 0.8 and kinds["ins"] > 300 and kinds["del"] > 300
    for l in (0, 8):
        assert ctx.vcf_transform(vcf, fasta, l) == o.vcf(vcf, fasta, l)
        assert ctx.vcf_tokenised_on_device()
    # odd shapes: no samples, one record, the last stride reaching the end of the reference
    for args in ((64, 1, 0), (1000, 62, 1), (16 * 7, 7, 3)):
        v, f = ctx.genvcf(*args, seed=5)
        assert ctx.vcf_transform(v, f, 0) == o.vcf(v, f, 0), args


def test_config3_full_size_vcf2eds(ctx):
    """vcf2eds: 1 Gb reference + 10 M records, 8 diploid samples (BASELINE configs[3]) against the oracle."""
    assert _free_hbm_gb() > 24, "needs an (almost) empty MI355X"
    vcf, fasta = ctx.genvcf(1_000_000_000, 10_000_000, 8, seed=42)
    assert len(vcf) > 600e6 and len(fasta) > 1.0e9
    got = ctx.vcf_transform(vcf, fasta, 0)
    assert ctx.vcf_tokenised_on_device()
    assert len(got[0]) + len(got[1]) > 1.3e9
    want = o.vcf(vcf, fasta, 0)
    assert got[2] == want[2]
    assert got[0] == want[0] and got[1] == want[1]


def test_config2_full_size_eds2leds_linear(ctx):
    """eds2leds LINEAR with .seds sources, genrandomeds 100 Mb reference at 10 % sites, l = 32 (BASELINE configs[2])."""
    assert _free_hbm_gb() > 24, "needs an (almost) empty MI355X"
    eds, seds, nsites = ctx.genrandomeds(100_000_000, 0.10, seed=42)
    assert nsites > 9_000_000 and len(eds) > 180e6
    got = ctx.leds_merge(eds, seds, 32, True)
    assert ctx.leds_tokenised_on_device()
    want = o.merge(eds, seds, 32, True)
    assert got[0] == want[0] and got[1] == want[1]
    st = ctx.eds_stats(got[0], got[1], 32)
    assert st["is_leds"] == 1


def test_shuffled_records_take_the_device_radix_sort(ctx):
    """A VCF whose records are in random order with pairwise distinct positions is ordered by the engine's own LSD radix
    sort on the device (k_rs_hist / k_rs_scatter: eight stable passes of eight bits on (POS, record index) pairs); the
    output must be the oracle's (the reference sorts with std::sort).  300 000 records of the configs[3] shape on a 30 Mb
    reference, positions beyond one radix tile and digit."""
    import random
    vcf, fasta = ctx.genvcf(30_000_000, 300_000, 8, seed=9)
    lines = vcf.split(b"\n")
    head = [x for x in lines if x.startswith(b"#")]
    body = [x for x in lines if x and not x.startswith(b"#")]
    assert len(body) == 300_000
    random.Random(3).shuffle(body)
    shuffled = b"\n".join(head + body) + b"\n"
    want = o.vcf(shuffled, fasta, 0)
    got = ctx.vcf_transform(shuffled, fasta, 0)
    assert got[0] == want[0] and got[1] == want[1]
    assert got[0] == ctx.vcf_transform(vcf, fasta, 0)[0]              # the same records in file order
