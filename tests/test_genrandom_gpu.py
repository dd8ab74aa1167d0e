"""GPU tests of the genrandomeds-shaped generator (edsx_genrandomeds, the genrandomeds tool): the text it writes has the
shape src/cpp/tools/genrandomeds.cpp:221-352 describes, parses with the oracle's EDS parser, and feeds the LINEAR merge."""
import os
import re
import subprocess

import pytest

import oracle_lib as o

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    import edsparser_amd
    c = edsparser_amd.Context(0)
    yield c
    c.close()


def _symbols(eds):
    return [s.split(b",") for s in re.findall(rb"\{([^}]*)\}", eds)]


def _sets(seds):
    return [[int(x) for x in s.split(b",")] for s in re.findall(rb"\{([^}]*)\}", seds)]


def test_shape_default_flags(ctx):
    bp, v = 2_000_000, 0.10
    eds, seds, nsites = ctx.genrandomeds(bp, v, seed=7)
    assert not eds.endswith(b"\n") and eds.startswith(b"{") and eds.endswith(b"}")
    syms, sets = _symbols(eds), _sets(seds)
    deg = [s for s in syms if len(s) > 1]
    assert len(deg) == nsites and abs(nsites - bp * v) < 5 * (bp * v * (1 - v)) ** 0.5      # binomial around bp * v
    assert sum(len(s) for s in syms) == len(sets)                                             # one source set per string
    assert all(2 <= len(s) <= 4 for s in deg)
    # the reference walks through every symbol once: common characters + one reference character per site = bp
    assert sum(len(s[0]) for s in syms if len(s) == 1) + len(deg) == bp
    assert set(eds) <= set(b"ACGT{},")
    # alternatives: first = one reference character; others SNP (1 char, different), insertion (ref + 1..10), deletion (empty)
    kinds = {"snp": 0, "ins": 0, "del": 0}
    for s in deg[:20000]:
        assert len(s[0]) == 1
        for a in s[1:]:
            if len(a) == 0:
                kinds["del"] += 1
            elif len(a) == 1:
                assert a != s[0]
                kinds["snp"] += 1
            else:
                assert a[:1] == s[0] and 2 <= len(a) <= 11
                kinds["ins"] += 1
    tot = sum(kinds.values())
    assert abs(kinds["snp"] / tot - 0.7) < 0.02 and abs(kinds["ins"] / tot - 0.15) < 0.02
    # sources: {0} for common blocks; per site the sets partition paths 1..4, string a holds path a + 1
    i = 0
    for s in syms[:5000]:
        if len(s) == 1:
            assert sets[i] == [0]
        else:
            part = sets[i:i + len(s)]
            assert sorted(x for p in part for x in p) == [1, 2, 3, 4]
            assert all(a + 1 in part[a] and part[a] == sorted(part[a]) for a in range(len(s)))
        i += len(s)
    # same seed, same text; another seed, another text
    assert ctx.genrandomeds(bp, v, seed=7)[0] == eds and ctx.genrandomeds(bp, v, seed=8)[0] != eds


def test_flags_and_consumers(ctx):
    eds, seds, nsites = ctx.genrandomeds(300_000, 0.05, min_alt=3, max_alt=6, var_len_max=4, snp_ratio=0.2, alphabet="ACGTN", seed=3)
    syms, sets = _symbols(eds), _sets(seds)
    deg = [s for s in syms if len(s) > 1]
    assert all(3 <= len(s) <= 6 for s in deg) and max(len(a) for s in deg for a in s) <= 5 and set(eds) <= set(b"ACGTN{},")
    assert max(x for st in sets for x in st) == 6                                             # max(max_alt, 3) paths
    # the generator's output is what the statistics and the merge take
    st = ctx.eds_stats(eds, seds, 0)
    assert st["num_degenerate_symbols"] == nsites and st["num_paths"] == 7 and st == o.eds_stats(eds, seds, 0)
    assert ctx.leds_merge(eds, seds, 8, True) == o.merge(eds, seds, 8, True)                  # LINEAR: every path takes one alternative per site
    # min-context mode: one site per segment (genrandomeds.cpp:86-108)
    e2, s2, n2 = ctx.genrandomeds(1_000_000, 0.02, min_context=30, seed=5)
    d2 = [s for s in _symbols(e2) if len(s) > 1]
    assert n2 == len(d2) == 20000
    assert ctx.eds_stats(e2, s2, 0) == o.eds_stats(e2, s2, 0)
    import edsparser_amd
    with pytest.raises(edsparser_amd.EdsxError) as ei:
        ctx.genrandomeds(1000, 0.1, min_alt=1)
    assert ei.value.code == 3 and "Minimum alternatives must be at least 2" in ei.value.message


def test_genrandomeds_cli(tmp_path):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "edsparser_amd", "host")], check=True)
    exe = os.path.join(ROOT, "edsparser_amd", "host", "build", "genrandomeds")
    out = tmp_path / "r.eds"
    r = subprocess.run([exe, "--ref-size-mb", "1", "-v", "0.05", "--seed", "11", "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Generating random EDS:" in r.stderr and "Reference size: 1 MB (1000000 bp)" in r.stderr
    assert "Sources written to:" in r.stderr and "[Performance] Runtime:" in r.stderr
    eds, seds = out.read_bytes(), (tmp_path / "r.seds").read_bytes()
    assert sum(len(s) for s in _symbols(eds)) == len(_sets(seds))
    r2 = subprocess.run([exe, "--ref-size-mb", "1", "-v", "0.05", "--seed", "11", "-o", str(tmp_path / "q.leds")], capture_output=True, text=True)
    assert r2.returncode == 0 and (tmp_path / "q.leds").read_bytes() == eds and (tmp_path / "q.seds").read_bytes() == seds
    r3 = subprocess.run([exe, "--ref-size-mb", "1", "--min-alternatives", "1", "-o", str(out)], capture_output=True, text=True)
    assert r3.returncode == 1 and "Error: Minimum alternatives must be at least 2" in r3.stderr
    r4 = subprocess.run([exe, "-o", str(out)], capture_output=True, text=True)
    assert r4.returncode == 1 and "the option '--ref-size-mb' is required but missing" in r4.stderr
