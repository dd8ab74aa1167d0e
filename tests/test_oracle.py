"""CPU tests: the oracle restatement against every golden vector the reference holds for the
hot path (tests/cpp/test_msa.cpp strings, data/ goldens), the SURVEY KATs and the fixtures
generated from the real reference library (tests/golden/make_golden.py)."""
import json
import os
import sys

import pytest

import oracle_lib as o
from conftest import GOLDEN


def _load(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def _rd(rel):
    return open(os.path.join(GOLDEN, rel), "rb").read()


@pytest.mark.parametrize("case", _load("kat_msa.json")["cases"], ids=lambda c: c["name"])
def test_msa_kat(case):
    msa = case["msa"].encode() if "msa" in case else _rd(case["msa_file"])
    eds, seds = o.msa(msa, case["l"])
    assert eds.decode() == case["eds"]
    assert seds.decode() == case["seds"]


def test_msa_rejects_single_sequence():
    with pytest.raises(o.OracleError):
        o.msa(b">a\nACGT\n", 0)


def test_msa_no_trailing_newline_and_shorter_row():
    # a later, shorter sequence leaves the tail common
    eds, seds = o.msa(b">a\nACGT\n>b\nAC\n", 0)
    assert (eds, seds) == (b"{ACGT}", b"{0}")
    # quirk (6): no trailing newline
    eds, seds = o.msa(b">a\nACGT\n>b\nACGA", 0)
    assert (eds, seds) == (b"{ACG}{T,A}", b"{0}{1}{2}")


@pytest.mark.parametrize("case", _load("kat_merge.json")["cases"], ids=lambda c: c["name"])
def test_merge_kat(case):
    seds = case["seds"].encode() if "seds" in case else None
    if "error" in case:
        with pytest.raises(o.OracleError) as ei:
            o.merge(case["eds"].encode(), seds, case["l"])
        assert str(ei.value) == case["error"]
        return
    out, so = o.merge(case["eds"].encode(), seds, case["l"])
    assert out.decode() == case["out"]
    if "seds_out" in case:
        assert so.decode() == case["seds_out"]


REF_EDS_GOLDENS = [("simple", 5), ("test_adjacent_degenerate", 1), ("test_adjacent_internal", 1),
                   ("test_degenerate_at_end", 4), ("test_iterative", 4), ("test_linear_sources", 4),
                   ("test_short_common", 4), ("test_short_edges", 4), ("test_short_first_common", 4)]


@pytest.mark.parametrize("name,l", REF_EDS_GOLDENS)
def test_merge_reference_data_goldens(name, l):
    """data/eds/X.eds -> data/eds/X_l<N>.eds (eds2leds CARTESIAN, compact) — reference goldens."""
    out, _ = o.merge(_rd("ref_data/eds/%s.eds" % name), None, l, True)
    assert out == _rd("ref_data/eds/%s_l%d.eds" % (name, l))


def test_merge_l0_rejected():
    with pytest.raises(o.OracleError) as ei:
        o.merge(b"{A}", None, 0)
    assert ei.value.code == 3


def test_merge_generated_goldens():
    cases = _load("gen_merge.json")["cases"]
    assert len(cases) >= 300
    for c in cases:
        seds = c["seds"].encode() if c["seds"] is not None else None
        try:
            out, so = o.merge(c["eds"].encode(), seds, c["l"], c["compact"])
            got = {"out": out.decode(), "seds_out": so.decode()}
        except o.OracleError as ex:
            got = {"error": str(ex)}
        assert got == c["expect"], c


@pytest.mark.parametrize("case", _load("kat_vcf.json")["cases"], ids=lambda c: c["name"])
def test_vcf_kat(case):
    k = _load("kat_vcf.json")
    v = case["vcf"].encode() if "vcf" in case else _rd(case["vcf_file"])
    f = _rd(case["fasta_file"]) if "fasta_file" in case else k["fasta_default"].encode()
    ee = case["eds"].encode() if "eds" in case else _rd(case["eds_file"])
    ss = case["seds"].encode() if "seds" in case else _rd(case["seds_file"])
    e, s, _ = o.vcf(v, f, case["l"])
    assert e == ee
    assert s == ss


def test_vcf_generated_goldens():
    cases = _load("gen_vcf.json")["cases"]
    assert len(cases) >= 200
    for c in cases:
        try:
            e, s, st = o.vcf(c["vcf"].encode(), c["fasta"].encode(), c["l"])
            got = {"eds": e.decode(), "seds": s.decode(), "stats": st}
        except o.OracleError as ex:
            got = {"error": str(ex)}
        assert got == c["expect"], c


@pytest.mark.skipif(not o.have_ref(), reason="oracle/_ref not built (reference absent on this box)")
def test_generated_goldens_still_match_reference():
    """In the build container: the committed fixtures are what the real reference produces now."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py"), "--check"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


# ---- round-2 fixtures (tests/golden/make_golden2.py): equal-POS VCFs of 17..200 records, quirk 22-27 shapes, merges of
# 500..2000 symbols with 8 rounds (LINEAR with 1 and 3 threads, CARTESIAN), BASELINE configs[0]'s shape at 0.2 MB
def _matches(packed, b):
    import hashlib
    if "text" in packed:
        return packed["text"].encode() == b
    return packed["len"] == len(b) and packed["sha256"] == hashlib.sha256(b).hexdigest()


def test_vcf_generated_goldens_round2():
    cases = _load("gen2_vcf.json")["cases"]
    assert sum(c["name"].startswith("samepos") for c in cases) >= 60
    for c in cases:
        e, s, st = o.vcf(c["vcf"].encode(), c["fasta"].encode(), c["l"])
        assert {"eds": e.decode(), "seds": s.decode(), "stats": st} == c["expect"], c["name"]


def test_merge_generated_goldens_round2():
    doc = _load("gen2_merge.json")
    for c in doc["cases"]:
        assert c["rounds"] >= 6
        out, so = o.merge(c["eds"].encode(), c["seds"].encode() if c["seds"] is not None else None, c["l"], c["compact"])
        assert _matches(c["expect"]["out"], out) and _matches(c["expect"]["seds_out"], so), c["name"]
        if "expect_threads3" in c:
            assert c["expect_threads3"] == c["expect"], c["name"]


def test_merge_baseline_config0_shape():
    """BASELINE configs[0]: genrandomeds @5 % -> eds2leds -l 10 (CARTESIAN, compact) on a 0.2 MB reference."""
    import hashlib
    from merge_cases import genrandomeds_shaped
    c0 = _load("gen2_merge.json")["configs0"]
    g = c0["generator"]
    eds, _ = genrandomeds_shaped(g["ref_mb"], g["v"], g["seed"])
    assert len(eds) == c0["eds_len"] and hashlib.sha256(eds).hexdigest() == c0["eds_sha256"]
    out, so = o.merge(eds, None, c0["l"], c0["compact"])
    assert _matches(c0["expect"]["out"], out) and _matches(c0["expect"]["seds_out"], so)


def test_eds_statistics_and_is_leds_generated_goldens():
    """EDS::Statistics + is_leds of ~800 inputs / outputs: the oracle restatement against the reference's numbers."""
    sys.path.insert(0, GOLDEN)
    from make_golden2 import stats_case_run
    gm, g2 = _load("gen_merge.json")["cases"], _load("gen2_merge.json")["cases"]
    cases = _load("gen2_stats.json")["cases"]
    assert len(cases) >= 700 and any(c["stats"].get("is_leds") == 0 for c in cases)
    for sc in cases:
        assert stats_case_run(o.eds_stats, sc, gm, g2) == sc["stats"], sc


@pytest.mark.skipif(not o.have_ref(), reason="oracle/_ref not built (reference absent on this box)")
def test_generated_goldens_round2_still_match_reference():
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden2.py"), "--check"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
