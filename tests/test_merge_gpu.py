"""GPU parity tests for the EDS -> l-EDS merge (edsx_leds_merge): reference goldens, SURVEY KATs,
fixtures generated from the real reference library, and random cases against the oracle."""
import json
import os
import random

import pytest

import oracle_lib as o
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import edsparser_amd
    c = edsparser_amd.Context(0)
    yield c
    c.close()


def _run(ctx, eds, seds, l, compact=True):
    import edsparser_amd
    try:
        out, so = ctx.leds_merge(eds, seds, l, compact)
        return {"out": out.decode(), "seds_out": so.decode()}
    except edsparser_amd.EdsxError as ex:
        return {"error": ex.message, "code": ex.code}


def test_kat(ctx):
    for c in json.load(open(os.path.join(GOLDEN, "kat_merge.json")))["cases"]:
        got = _run(ctx, c["eds"].encode(), c["seds"].encode() if "seds" in c else None, c["l"])
        if "error" in c:
            assert got.get("error") == c["error"], c["name"]
        else:
            assert got["out"] == c["out"], c["name"]
            if "seds_out" in c:
                assert got["seds_out"] == c["seds_out"], c["name"]


@pytest.mark.parametrize("name,l", [("simple", 5), ("test_adjacent_degenerate", 1), ("test_adjacent_internal", 1),
                                    ("test_degenerate_at_end", 4), ("test_iterative", 4), ("test_linear_sources", 4),
                                    ("test_short_common", 4), ("test_short_edges", 4), ("test_short_first_common", 4)])
def test_reference_data_goldens(ctx, name, l):
    eds = open(os.path.join(GOLDEN, "ref_data/eds/%s.eds" % name), "rb").read()
    want = open(os.path.join(GOLDEN, "ref_data/eds/%s_l%d.eds" % (name, l)), "rb").read()
    assert _run(ctx, eds, None, l)["out"].encode() == want


def test_generated_reference_fixtures(ctx):
    cases = json.load(open(os.path.join(GOLDEN, "gen_merge.json")))["cases"]
    for c in cases:
        got = _run(ctx, c["eds"].encode(), c["seds"].encode() if c["seds"] is not None else None, c["l"], c["compact"])
        got.pop("code", None)
        assert got == c["expect"], c


def test_l0_is_invalid_argument(ctx):
    got = _run(ctx, b"{A}", None, 0)
    assert got["code"] == 3


def _random_eds(rng, n, paths, with_src, p_deg=0.45):
    syms, srcs = [], []
    for _ in range(n):
        k = 1 if rng.random() >= p_deg else rng.randint(2, 4)
        strs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(0, 12))) for _ in range(k)]
        syms.append(strs)
        if k == 1:
            srcs.append([0])
        else:
            # every path takes exactly one alternative (genrandomeds-style), so LINEAR never empties
            choice = [rng.randrange(k) for _ in range(paths)]
            for a in range(k):
                ids = [p + 1 for p in range(paths) if choice[p] == a]
                srcs.append(ids if ids else [rng.randint(1, paths)])
    text = "".join("{" + ",".join(s) + "}" for s in syms)
    seds = "".join("{" + ",".join(map(str, s)) + "}" for s in srcs) if with_src else None
    return text.encode(), (seds.encode() if seds else None)


@pytest.mark.parametrize("with_src", [False, True])
def test_random_larger_inputs_vs_oracle(ctx, with_src):
    rng = random.Random(99 + with_src)
    for it in range(12):
        n = rng.choice([50, 300, 2000]) if with_src else rng.choice([40, 200, 1000])
        eds, seds = _random_eds(rng, n, paths=rng.randint(2, 70), with_src=with_src, p_deg=0.45 if with_src else 0.2)
        l = rng.choice([1, 3, 8, 20]) if with_src else rng.choice([1, 2, 3])
        for compact in (True, False):
            try:
                want = o.merge(eds, seds, l, compact)
                want = {"out": want[0].decode(), "seds_out": want[1].decode()}
            except o.OracleError as ex:
                want = {"error": str(ex)}
            got = _run(ctx, eds, seds, l, compact)
            got.pop("code", None)
            assert got == want, (it, n, l, compact)
