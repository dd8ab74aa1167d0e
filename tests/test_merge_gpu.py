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


def _matches(packed, b):
    import hashlib
    if "text" in packed:
        return packed["text"].encode() == b
    return packed["len"] == len(b) and packed["sha256"] == hashlib.sha256(b).hexdigest()


def test_generated_reference_fixtures_round2(ctx):
    """500..2000 symbols, 8 merge rounds, LINEAR and CARTESIAN (tests/golden/make_golden2.py ran the reference)."""
    for c in json.load(open(os.path.join(GOLDEN, "gen2_merge.json")))["cases"]:
        out, so = ctx.leds_merge(c["eds"].encode(), c["seds"].encode() if c["seds"] is not None else None, c["l"], c["compact"])
        assert _matches(c["expect"]["out"], out) and _matches(c["expect"]["seds_out"], so), c["name"]


def test_baseline_config0_shape(ctx):
    """BASELINE configs[0]: genrandomeds @5 % -> eds2leds -l 10 (CARTESIAN, compact), 0.2 MB reference, 28 MB out;
    expected bytes = the reference's (length + SHA-256 in the fixture)."""
    import hashlib
    from merge_cases import genrandomeds_shaped
    c0 = json.load(open(os.path.join(GOLDEN, "gen2_merge.json")))["configs0"]
    g = c0["generator"]
    eds, _ = genrandomeds_shaped(g["ref_mb"], g["v"], g["seed"])
    assert len(eds) == c0["eds_len"] and hashlib.sha256(eds).hexdigest() == c0["eds_sha256"]
    out, so = ctx.leds_merge(eds, None, c0["l"], c0["compact"])
    assert _matches(c0["expect"]["out"], out) and _matches(c0["expect"]["seds_out"], so)


def test_eds_statistics_and_is_leds(ctx):
    """edsx_eds_stats (device reductions) against the reference's EDS::Statistics / is_leds numbers for ~800 texts
    (inputs and outputs of the merge fixtures), error texts included."""
    import sys
    sys.path.insert(0, GOLDEN)
    from make_golden2 import stats_case_run
    gm = json.load(open(os.path.join(GOLDEN, "gen_merge.json")))["cases"]
    g2 = json.load(open(os.path.join(GOLDEN, "gen2_merge.json")))["cases"]
    for sc in json.load(open(os.path.join(GOLDEN, "gen2_stats.json")))["cases"]:
        assert stats_case_run(ctx.eds_stats, sc, gm, g2) == sc["stats"], sc


def test_merge_outputs_validate_as_leds(ctx):
    """Self-check: what edsx_leds_merge writes is an l-EDS by the device validator, for random larger inputs."""
    rng = random.Random(99)
    for _ in range(40):
        linear = rng.random() < 0.6
        eds, seds = _random_eds(rng, rng.randint(2, 400) if linear else rng.randint(2, 14), 6, linear)   # (CARTESIAN products explode)
        l = rng.choice([1, 2, 4, 8, 16])
        got = _run(ctx, eds, seds, l, False)          # FULL brackets: COMPACT text drops an empty single-string symbol
        if "error" in got:
            continue
        st = ctx.eds_stats(got["out"].encode(), got["seds_out"].encode() if linear else None, l)
        assert st["is_leds"] == 1, (eds, l)
        assert st == o.eds_stats(got["out"].encode(), got["seds_out"].encode() if linear else None, l)


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


def _big_eds(rng, nsym, paths, compact_in, spacing):
    """genrandomeds-like text of >= 1 MB: long common blocks, sites `spacing` symbols apart on average.
    compact_in: non-degenerate symbols without braces (EDS::save COMPACT), plus whitespace noise."""
    eds, seds = [], []
    for i in range(nsym):
        if i % 2 == 0:
            s = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, spacing)))
            eds.append(s if compact_in else "{" + s + "}")
            seds.append("{0}")
        else:
            k = rng.randint(2, 4)
            alts = ["".join(rng.choice("ACGT") for _ in range(rng.randint(0, 3))) for _ in range(k)]
            choice = [p if p < k else rng.randrange(k) for p in range(paths)]
            eds.append("{" + ",".join(alts) + "}")
            for a in range(k):
                seds.append("{" + ",".join(str(p + 1) for p in range(paths) if choice[p] == a) + "}")
        if compact_in and i % 97 == 0:
            eds.append("\n" if i % 2 else " \t")
            seds.append("\n")
    return "".join(eds).encode(), "".join(seds).encode()


def _long_leaf_eds(rng, nsym, paths, lens):
    """variant sites between common strings whose lengths are drawn from `lens` (long leaves of the merge trees)"""
    eds, seds = [], []
    for i in range(nsym):
        if i % 2 == 0:
            n = rng.choice(lens)
            eds.append("{" + "".join(rng.choice("ACGT") for _ in range(n)) + "}")
            seds.append("{0}")
        else:
            k = rng.randint(2, 3)
            alts = ["".join(rng.choice("ACGT") for _ in range(rng.randint(0, 2))) for _ in range(k)]
            choice = [q if q < k else rng.randrange(k) for q in range(paths)]
            eds.append("{" + ",".join(alts) + "}")
            for a in range(k):
                seds.append("{" + ",".join(str(q + 1) for q in range(paths) if choice[q] == a) + "}")
    return "".join(eds).encode(), "".join(seds).encode()


@pytest.mark.gpu
@pytest.mark.parametrize("lens,nsym", [((1, 5, 255, 256, 257, 300), 400), ((3, 4095, 4096, 4113, 9000), 120),
                                       ((70000, 2, 17), 40), ((1, 2), 6000), ((1, 2), 40000)])
def test_final_text_long_leaves_and_deep_trees(ctx, lens, nsym):
    """k_fin_write expands the merge trees level by level: leaves of 256 bytes and more are copied by the whole
    workgroup (several 4 KB trips for the longest), short ones by the thread that pops them; the strings per
    workgroup follow the mean string length (1 .. 256).  Context lengths below, between and above the leaf lengths
    give single-leaf strings, mixed trees and one tree over the whole text; LINEAR and CARTESIAN, both bracket modes."""
    rng = random.Random(len(lens) * 1000 + nsym)
    eds, seds = _long_leaf_eds(rng, nsym, 5, lens)
    for sd in (seds, None):
        for l in (1, 6, 280, 5000, 100000):
            if sd is None and l > 1:
                continue                                     # CARTESIAN products of long chains explode
            for compact in (True, False):
                try:
                    want = o.merge(eds, sd, l, compact)
                    want = {"out": want[0].decode(), "seds_out": want[1].decode()}
                except o.OracleError as ex:
                    want = {"error": str(ex)}
                got = _run(ctx, eds, sd, l, compact)
                got.pop("code", None)
                assert got == want, (lens, nsym, sd is None, l, compact)


@pytest.mark.parametrize("compact_in", [False, True])
def test_parallel_tokenisers_large_inputs(ctx, compact_in):
    """Inputs of 1 MB and more are cut behind '}' and tokenised by several host threads; the result
    must be what the sequential tokeniser gives (oracle), for FULL and COMPACT input text, LINEAR and
    CARTESIAN; a malformed large input must still produce the reference's error text."""
    rng = random.Random(7 + compact_in)
    eds, seds = _big_eds(rng, 140000, 6, compact_in, 60)
    assert len(eds) >= 1 << 20 and len(seds) >= 1 << 20
    for sd, l in ((seds, 24), (None, 2)):
        want = o.merge(eds, sd, l, True)
        got = _run(ctx, eds, sd, l, True)
        assert got.get("out") == want[0].decode() and got.get("seds_out") == want[1].decode(), (compact_in, l)
    bad = eds[:700000] + b"{A{C}" + eds[700000:]
    for sd in (seds, None):
        try:
            want = o.merge(bad, sd, 5, True)
            want = {"out": want[0].decode(), "seds_out": want[1].decode()}
        except o.OracleError as ex:
            want = {"error": str(ex)}
        got = _run(ctx, bad, sd, 5, True)
        got.pop("code", None)
        assert got == want
    bad_s = seds[:100000] + b"{x}" + seds[100000:]
    try:
        o.merge(eds, bad_s, 5, True)
        raise AssertionError("oracle accepted a malformed sEDS")
    except o.OracleError as ex:
        got = _run(ctx, eds, bad_s, 5, True)
        assert got.get("error") == str(ex)


def test_parallel_tokenisers_fuzz(ctx):
    """Random damage to large .eds/.seds inputs (bytes inserted, deleted or replaced anywhere): whatever
    the chunk-parallel tokenisers make of it, result or error text must equal the oracle's (= the
    reference's sequential parser)."""
    rng = random.Random(2024)
    eds0, seds0 = _big_eds(rng, 140000, 5, False, 40)
    junk = b"{},AC \n0123x{{}}"
    for it in range(24):
        eds, seds = bytearray(eds0), bytearray(seds0)
        target = eds if it % 2 == 0 else seds
        for _ in range(rng.randint(1, 3)):
            pos = rng.randrange(len(target))
            op = rng.random()
            if op < 0.4:
                target[pos:pos] = bytes([rng.choice(junk)])
            elif op < 0.7:
                del target[pos]
            else:
                target[pos] = rng.choice(junk)
        eds, seds = bytes(eds), bytes(seds)
        sd = seds if it % 3 else None
        try:
            want = o.merge(eds, sd, 6, True)
            want = {"out": want[0].decode(), "seds_out": want[1].decode()}
        except o.OracleError as ex:
            want = {"error": str(ex)}
        got = _run(ctx, eds, sd, 6, True)
        got.pop("code", None)
        assert got == want, it


def test_randomized_campaign(ctx):
    """400 inputs of the randomized campaign (tests/merge_cases.py), results and error texts against the
    oracle; 2800 further cases (seeds 1-4 x 700) were run once on the MI355X box without a mismatch."""
    from merge_cases import campaign_eds
    rng = random.Random(11)
    ran = 0
    for it in range(400):
        eds, seds, l, compact, desc = campaign_eds(rng)
        if desc[-1] > 100000:
            continue
        ran += 1
        try:
            w = o.merge(eds, seds, l, compact)
            want = {"out": w[0].decode(), "seds_out": w[1].decode()}
        except o.OracleError as ex:
            want = {"error": str(ex)}
        got = _run(ctx, eds, seds, l, compact)
        got.pop("code", None)
        assert got == want, (it, desc, l, compact)
    assert ran > 300


def test_device_tokeniser_is_taken_and_equals_the_host_tokenisers(ctx):
    """Plain text (what the tools write) is tokenised on the GPU; text the kernels cannot place byte for byte goes to
    the host tokenisers.  Both ways must give the same bytes (EDSX_HOST_TOKENIZER=1 forces the host path), and which
    way was taken is visible through edsx_leds_tokenised_on_device."""
    from merge_cases import campaign_eds
    rng = random.Random(5150)
    on_dev = on_host = 0
    for it in range(250):
        eds, seds, l, compact, desc = campaign_eds(rng)
        if desc[-1] > 100000:
            continue
        got = _run(ctx, eds, seds, l, compact)
        dev = ctx.leds_tokenised_on_device()
        os.environ["EDSX_HOST_TOKENIZER"] = "1"
        try:
            host = _run(ctx, eds, seds, l, compact)
            assert not ctx.leds_tokenised_on_device()
        finally:
            del os.environ["EDSX_HOST_TOKENIZER"]
        assert got == host, (it, desc)
        plain = not any(ch in eds[:len(eds.rstrip())] for ch in b" \n\t\r") and "error" not in got
        if plain and eds.strip():
            assert dev, (it, eds[:80])
        on_dev += dev
        on_host += not dev
    assert on_dev > 100 and on_host > 10


def test_device_tokeniser_hands_odd_text_to_the_host(ctx):
    cases = [
        (b"{A,C} GG{T}", None),              # inner whitespace
        (b"A,C{G,T}AAAA", None),             # comma outside braces: the bare run becomes a degenerate symbol
        (b"{A{C}}", None),                   # nesting
        (b"{A,C}}", None),                   # stray brace
        (b"{A,C", None),                     # unterminated
        (b"{A,C}{G}", b"{1}{2}"),            # source count differs from the cardinality
        (b"{A,C}{G}", b"{1}{}{0}"),          # empty path set
        (b"{A,C}{G}", b"{1}{,,}{0}"),        # ... of commas only
        (b"{A,C}{G}", b"{1}{2}}{0}"),        # stray brace in the sources
        (b"{A,C}{G}", b"{1}{2}{{0}"),
        (b"{A,C}{G}", b"{1}{2x}{0}"),        # character that does not belong
        (b"{A,C}{G}", b"{1}{99999999999}{0}"),
        (b"", None), (b"\n", None),
    ]
    for eds, seds in cases:
        try:
            w = o.merge(eds, seds, 2, True)
            want = {"out": w[0].decode(), "seds_out": w[1].decode()}
        except o.OracleError as ex:
            want = {"error": str(ex)}
        got = _run(ctx, eds, seds, 2, True)
        got.pop("code", None)
        assert got == want, (eds, seds)
        assert not ctx.leds_tokenised_on_device(), (eds, seds)
    # plain text, all spellings: on the device
    for eds, seds in [(b"{A,C}GGGG{T,}\n", None), (b"ACGT", None), (b"{}{A}{,}", None), (b"{A,C}{G}", b"{1,2}{0,3}{0}\n"),
                      (b"{AC}GT{A,C}", b"{0}{0}{007}{1,,2}")]:
        try:
            w = o.merge(eds, seds, 3, False)
            want = {"out": w[0].decode(), "seds_out": w[1].decode()}
        except o.OracleError as ex:
            want = {"error": str(ex)}
        got = _run(ctx, eds, seds, 3, False)
        got.pop("code", None)
        assert got == want, (eds, seds)
        assert ctx.leds_tokenised_on_device(), (eds, seds)


def test_device_tokeniser_block_and_thread_boundaries(ctx):
    """The device tokenisers work on 4 KB blocks of 256 x 16 bytes: every brace, comma, bare-run start and id of a text is
    moved across those boundaries (a bare prefix of p letters in the .eds, an id with p leading zeros in the .seds shift
    everything behind them by p), and the result is compared with the host tokenisers and the oracle."""
    rng = random.Random(4096)
    syms = []
    for _ in range(900):
        k = rng.choice([1, 1, 2, 3, 4])
        strs = ["".join(rng.choice("ACGT") for _ in range(rng.choice([0, 1, 2, 5, 17]))) for _ in range(k)]
        if k == 1 and not strs[0]:
            strs[0] = "A"
        syms.append(strs)
    body = "".join("{" + ",".join(x) + "}" for x in syms)
    # (the first string of every symbol is on all paths, so no merge comes out empty)
    sbody = "".join("{" + (",".join(str(rng.randint(1, 1500)) for _ in range(rng.randint(1, 3))) if j else "0") + "}"
                    for x in syms for j in range(len(x)))
    assert len(body) > 3 * 4096 and len(sbody) > 3 * 4096
    shifts = list(range(1, 18)) + [31, 32, 33] + list(range(4085, 4108, 2)) + [8191, 8192, 8193]

    def both_tokenisers(fn):
        got = fn()
        assert ctx.leds_tokenised_on_device()
        os.environ["EDSX_HOST_TOKENIZER"] = "1"
        try:
            host = fn()
            assert not ctx.leds_tokenised_on_device()
        finally:
            del os.environ["EDSX_HOST_TOKENIZER"]
        return got, host

    for p_ in shifts:
        eds = ("A" * p_ + body).encode()
        seds = ("{" + "0" * p_ + "}" + sbody).encode()
        got, host = both_tokenisers(lambda: _run(ctx, eds, seds, 4, True))
        assert got == host, p_
        # (without sources the merge is CARTESIAN and its output explodes: the tokenisers alone, through the statistics call)
        got, host = both_tokenisers(lambda: ctx.eds_stats(eds + b"\n", None, 4))
        assert got == host, p_
        if p_ in (1, 4097):
            try:
                w = o.merge(eds, seds, 4, True)
                want = {"out": w[0].decode(), "seds_out": w[1].decode()}
            except o.OracleError as ex:
                want = {"error": str(ex)}
            g = _run(ctx, eds, seds, 4, True)
            g.pop("code", None)
            assert g == want, p_
    # a text that is rejected at the very end (after the arrays were filled) still takes the host path
    import edsparser_amd
    with pytest.raises(edsparser_amd.EdsxError):
        ctx.eds_stats((body + "{A").encode(), None, 4)
    assert not ctx.leds_tokenised_on_device()


def test_deep_tree_fallback_paths():
    """Final text when the merge trees do not fit the stacks: k_fin_write's LDS stack overflows -> the batch goes to the
    serial walk -> after more rounds than its own stack holds, that walk uses a per-string stack in HBM sized by the number
    of rounds.  Real inputs need trees deeper than 96 for this; `libedsx_smallstacks.so` (built by edsparser_amd.build for
    the tests only: 256-entry LDS stack, 2-entry register stack) takes ordinary campaign inputs down that path.  A child
    process loads it through EDSX_LIB, so the suite's own library stays the product build."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "edsparser_amd", "libedsx_smallstacks.so")
    assert os.path.exists(lib), "python -m edsparser_amd.build builds it"
    code = """
import random, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import edsparser_amd, oracle_lib as o
from merge_cases import campaign_eds
ctx = edsparser_amd.Context(0)
rng = random.Random(96)
ran = merged = 0
for it in range(220):
    eds, seds, l, compact, desc = campaign_eds(rng)
    if desc[-1] > 100000:
        continue
    try:
        w = o.merge(eds, seds, l, compact)
        want = (w[0], w[1])
    except o.OracleError as ex:
        want = str(ex)
    try:
        got = ctx.leds_merge(eds, seds, l, compact)
    except edsparser_amd.EdsxError as ex:
        got = ex.message
    assert got == want, (it, desc, l, compact)
    ran += 1
    merged += isinstance(want, tuple) and want[0].strip() != eds.strip()
print("ran", ran, "merged", merged)
assert ran > 150 and merged > 60
""" % (root, os.path.join(root, "tests"))
    env = dict(os.environ, EDSX_LIB=lib)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_large_downloads_every_size_remainder(ctx):
    """Outputs of 16 MB and more come back through pinned chunks copied by eight host threads (PinnedDownload): the
    split must cover sizes with every remainder modulo the thread count and the 64-byte rounding (a first version
    dropped the last len % 8 bytes when len / 8 was a multiple of 64 — found by the full-size configs[2] run)."""
    import hashlib
    base = b"{A,C}" + b"G" * 59                      # 64 bytes per unit, merges to itself for l <= 59
    unit_n = (17 << 20) // 64
    for extra in range(0, 18):
        eds = base * unit_n + b"{A,C}" + b"T" * extra
        out, so = ctx.leds_merge(eds, None, 8, True)
        want = (base.replace(b"{A,C}", b"{A,C}") * unit_n + b"{A,C}" + b"T" * extra + b"\n")
        assert len(out) == len(want) and hashlib.sha256(out).digest() == hashlib.sha256(want).digest(), extra
        assert so == b""
