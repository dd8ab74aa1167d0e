"""GPU parity tests for VCF -> EDS / l-EDS (edsx_vcf_transform): the reference's data/vcf goldens,
SURVEY KATs, 300 fixtures generated from the real reference library, larger random cases vs the oracle."""
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


def _run(ctx, vcf, fasta, l):
    import edsparser_amd
    try:
        e, s, st = ctx.vcf_transform(vcf, fasta, l)
        return {"eds": e.decode(), "seds": s.decode(), "stats": st}
    except edsparser_amd.EdsxError as ex:
        return {"error": ex.message}


def _rd(rel):
    return open(os.path.join(GOLDEN, rel), "rb").read()


def test_kat_and_reference_goldens(ctx):
    k = json.load(open(os.path.join(GOLDEN, "kat_vcf.json")))
    for c in k["cases"]:
        v = c["vcf"].encode() if "vcf" in c else _rd(c["vcf_file"])
        f = _rd(c["fasta_file"]) if "fasta_file" in c else k["fasta_default"].encode()
        ee = c["eds"].encode() if "eds" in c else _rd(c["eds_file"])
        ss = c["seds"].encode() if "seds" in c else _rd(c["seds_file"])
        got = _run(ctx, v, f, c["l"])
        assert got.get("eds", "").encode() == ee, c["name"]
        assert got.get("seds", "").encode() == ss, c["name"]


def test_generated_reference_fixtures(ctx):
    cases = json.load(open(os.path.join(GOLDEN, "gen_vcf.json")))["cases"]
    for c in cases:
        got = _run(ctx, c["vcf"].encode(), c["fasta"].encode(), c["l"])
        assert got == c["expect"], c


def test_generated_reference_fixtures_round2(ctx):
    """17..200 records with many equal POS (the reference's unstable std::sort decides their order, quirk 29) and the
    SURVEY quirk 22-27 shapes; expected text from the reference itself (tests/golden/make_golden2.py)."""
    cases = json.load(open(os.path.join(GOLDEN, "gen2_vcf.json")))["cases"]
    for c in cases:
        got = _run(ctx, c["vcf"].encode(), c["fasta"].encode(), c["l"])
        assert got == c["expect"], c["name"]


def _random_vcf(rng, L, nvar, ns, lw):
    ref = "".join(rng.choice("ACGT") for _ in range(L))
    fasta = ">chr1 synthetic\n" + "\n".join(ref[i:i + lw] for i in range(0, L, lw)) + "\n"
    pos = sorted(rng.sample(range(1, L + 1), nvar))
    lines = ["##fileformat=VCFv4.2", "\t".join(["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"] +
                                                ["S%d" % i for i in range(ns)])]
    for p in pos:
        x = rng.random()
        if x < 0.7:
            r = ref[p - 1]
            alts = [rng.choice([b for b in "ACGT" if b != r])]
            if rng.random() < 0.1:
                alts.append(rng.choice("ACGT"))
        elif x < 0.85:
            r = ref[p - 1]
            alts = [r + "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 10)))]
        else:
            r = ref[p - 1:p + rng.randint(1, 10)]
            alts = [r[0]]
        gts = ["|".join(str(rng.randint(0, len(alts)) if rng.random() < 0.3 else 0) for _ in range(2)) for _ in range(ns)]
        lines.append("\t".join(["chr1", str(p), ".", r, ",".join(alts), ".", "PASS", ".", "GT"] + gts))
    return ("\n".join(lines) + "\n").encode(), fasta.encode()


@pytest.mark.parametrize("L,nvar,ns,lw,l", [(5000, 300, 8, 60, 0), (20000, 1500, 3, 70, 0), (3000, 200, 70, 3000, 0),
                                            (4000, 150, 4, 60, 6)])
def test_random_larger_inputs_vs_oracle(ctx, L, nvar, ns, lw, l):
    rng = random.Random(L + nvar)
    vcf, fasta = _random_vcf(rng, L, nvar, ns, lw)
    try:
        e, s, st = o.vcf(vcf, fasta, l)
        want = {"eds": e.decode(), "seds": s.decode(), "stats": st}
    except o.OracleError as ex:
        want = {"error": str(ex)}
    assert _run(ctx, vcf, fasta, l) == want


def test_multithreaded_tokeniser_large_vcf(ctx):
    """VCF files of 4 MB and more are cut at line starts and tokenised by several host threads;
    records, counters and group order must still equal the oracle's (which reads line by line).
    The file also carries malformed lines and unsupported structural variants across the cuts."""
    rng = random.Random(99)
    vcf, fasta = _random_vcf(rng, 400000, 40000, 24, 60)
    lines = vcf.decode().split("\n")
    for i in range(200, len(lines) - 1, 997):
        lines.insert(i, "chr1\tnot_a_number" if i % 2 else "chr1\t%d\t.\tA\t<INV>\t.\tPASS\t." % (i % 4000 + 1))
    vcf = "\n".join(lines).encode()
    assert len(vcf) >= 4 << 20
    e, s, st = o.vcf(vcf, fasta, 0)
    assert _run(ctx, vcf, fasta, 0) == {"eds": e.decode(), "seds": s.decode(), "stats": st}


def test_multithreaded_tokeniser_fuzz(ctx):
    """Random damage to a large VCF (tabs turned into spaces, fields removed, junk lines, odd genotypes,
    symbolic alleles, CR line ends): records, counters and output must equal the oracle's."""
    rng = random.Random(4242)
    vcf0, fasta = _random_vcf(rng, 300000, 36000, 24, 60)
    lines0 = vcf0.decode().split("\n")
    assert len(vcf0) >= 4 << 20
    for it in range(8):
        lines = list(lines0)
        for _ in range(400):
            i = rng.randrange(2, len(lines) - 1)
            f = lines[i].split("\t")
            op = rng.randrange(9)
            if op == 0:
                lines[i] = lines[i].replace("\t", " ")
            elif op == 1 and len(f) > 6:
                lines[i] = "\t".join(f[:rng.randint(1, 6)])
            elif op == 2:
                lines[i] = rng.choice(["", "#junk", "chr1", "\t\t\t", "chr1\t12\t.\tA"])
            elif op == 3 and len(f) > 10:
                f[9 + rng.randrange(len(f) - 9)] = rng.choice(["./.", ".", "1", "0/1/2", "x|1", "1|", "|", "0|1:9:x", "", "99999999999|0"])
                lines[i] = "\t".join(f)
            elif op == 4 and len(f) > 5:
                f[4] = rng.choice(["<DEL>", "<INS>", "<INV>", "A,<DEL>", "<DUP>,C", ",", "A,,C", "C,", "<>", "<", "."])
                lines[i] = "\t".join(f)
            elif op == 5:
                lines[i] = lines[i] + "\r"
            elif op == 6 and len(f) > 2:
                f[1] = rng.choice(["0", "-5", "+7", " 12", "12abc", "abc", "", "99999999999999999999999", "300001"])
                lines[i] = "\t".join(f)
            elif op == 7:
                lines[i] = lines[i].replace("\t", "\t\t", 2)
            else:
                lines.insert(i, lines[rng.randrange(2, len(lines) - 1)])
        vcf = "\n".join(lines).encode()
        for l in ((0, 6) if it < 2 else (0,)):
            try:
                e, s, st = o.vcf(vcf, fasta, l)
                want = {"eds": e.decode(), "seds": s.decode(), "stats": st}
            except o.OracleError as ex:
                want = {"error": str(ex)}
            assert _run(ctx, vcf, fasta, l) == want, (it, l)


def test_randomized_campaign(ctx):
    """80 random VCF/FASTA pairs: reference length 30..30000, site density up to 0.4 (long overlap chains),
    0..70 samples, several FASTA line widths, shuffled record order, l in {0, 3, 10}; 900 cases of the same
    generator were run once on the MI355X box without a mismatch."""
    rng = random.Random(21)
    for it in range(80):
        L = rng.choice([30, 100, 1000, 5000, 30000])
        nvar = max(1, min(L - 12, int(L * rng.choice([0.01, 0.05, 0.2, 0.4]))))
        ns = rng.choice([0, 1, 2, 8, 63, 64, 65, 70])
        lw = rng.choice([60, 70, 7, L, 61])
        l = rng.choice([0, 0, 0, 3, 10]) if nvar <= 60 else 0
        vcf, fasta = _random_vcf(rng, L, nvar, ns, lw)
        if rng.random() < 0.3:
            lines = vcf.decode().split("\n")
            head, body = lines[:2], [x for x in lines[2:] if x]
            rng.shuffle(body)
            vcf = ("\n".join(head + body) + "\n").encode()
        try:
            e, s, st = o.vcf(vcf, fasta, l)
            want = {"eds": e.decode(), "seds": s.decode(), "stats": st}
        except o.OracleError as ex:
            want = {"error": str(ex)}
        assert _run(ctx, vcf, fasta, l) == want, (it, L, nvar, ns, lw, l)


def _want(vcf, fasta, l=0):
    try:
        e, s, st = o.vcf(vcf, fasta, l)
        return {"eds": e.decode(), "seds": s.decode(), "stats": st}
    except o.OracleError as ex:
        return {"error": str(ex)}


def test_device_tokeniser_single_damage_per_file(ctx):
    """One oddity per small file, every kind the device tokeniser must either place exactly or hand to the host
    tokeniser: the result must equal the oracle's either way (a wrongly accepted line would show here, where no
    second oddity in the same file can trigger the fallback for it)."""
    rng = random.Random(808)
    gts = ["./.", ".", "1", "0/1/2", "x|1", "1|", "|", "0|1:9:x", ":", "99999999999|0", "0|1\r", "+1|0", " 1|0", "01|1", "1x|0", "|1", "0||1", "0/1|1"]
    alts = ["<DEL>", "<INS>", "<INV>", "A,<DEL>", "<DUP>,C", ",", "A,,C", "C,", "<>", "<", ".", "*", "<DEL>,<INS>", "ACGTACGT", "<DELX>", "<del>"]
    poss = ["0", "-5", "+7", " 12", "12abc", "abc", "99999999999999999999999", "18446744073709551615", "0012", "1", "3000"]
    wholes = ["", "#junk", "chr1", "\t\t\t", "chr1\t12\t.\tA", "chr1 14 . A G . PASS . GT 0|1 1|1", "chr1\t15\t.\tA\tG", "chr1\t16\t.\tA\tG\t.\tPASS\t.\tGT",
              "\tchr1\t17\t.\tA\tG\t.\tPASS\t.\tGT\t0|1\t0|0", "chr1\t18\t.\tA\tG\t.\tPASS\t.\tGT\t0|1\t", "chr1\t19\t.\tA\tG\t\t.\tPASS\t.\tGT\t0|1"]
    taken = {True: 0, False: 0}
    n = 0
    for kind, choices in (("gt", gts), ("alt", alts), ("pos", poss), ("line", wholes), ("cr", ["\r"]), ("tabs", ["x"])):
        for ch in choices:
            for ns in (0, 2):
                vcf, fasta = _random_vcf(rng, 3000, 25, ns, 60)
                lines = vcf.decode().split("\n")
                i = rng.randrange(2, len(lines) - 1)
                f = lines[i].split("\t")
                if kind == "gt":
                    if ns == 0:
                        continue
                    f[9 + rng.randrange(ns)] = ch
                    lines[i] = "\t".join(f)
                elif kind == "alt":
                    f[4] = ch
                    lines[i] = "\t".join(f)
                elif kind == "pos":
                    f[1] = ch
                    lines[i] = "\t".join(f)
                elif kind == "line":
                    lines.insert(i, ch)
                elif kind == "cr":
                    lines[i] += "\r"
                else:
                    lines[i] = lines[i].replace("\t", "\t\t", 1)
                for tail in ("\n", ""):
                    v = ("\n".join(lines[:-1]) + tail).encode()
                    got = _run(ctx, v, fasta, 0)
                    assert got == _want(v, fasta), (kind, ch, ns, tail, lines[i])
                    taken[ctx.vcf_tokenised_on_device()] += 1
                    n += 1
    assert n > 150 and taken[True] > 20 and taken[False] > 60, taken


def test_device_tokeniser_is_taken_and_equals_the_host_tokeniser(ctx):
    """Plain files go through the device tokeniser; EDSX_HOST_TOKENIZER=1 forces the host tokeniser: same bytes, same
    counters; shuffled files and duplicate positions included (the sort is the host's in both)."""
    rng = random.Random(909)
    for it in range(40):
        L = rng.choice([100, 1000, 5000, 30000])
        nvar = max(1, min(L - 12, int(L * rng.choice([0.01, 0.05, 0.2, 0.4]))))
        ns = rng.choice([0, 1, 2, 8, 64, 70])
        vcf, fasta = _random_vcf(rng, L, nvar, ns, rng.choice([60, 7, L]))
        lines = vcf.decode().split("\n")
        head, body = lines[:2], [x for x in lines[2:] if x]
        if it % 3 == 1:
            rng.shuffle(body)
        if it % 3 == 2:
            body += [rng.choice(body) for _ in range(len(body) // 2)]
            rng.shuffle(body)
        vcf = ("\n".join(head + body) + ("\n" if it % 2 else "")).encode()
        got = _run(ctx, vcf, fasta, 0)
        assert ctx.vcf_tokenised_on_device(), it
        os.environ["EDSX_HOST_TOKENIZER"] = "1"
        try:
            host = _run(ctx, vcf, fasta, 0)
            assert not ctx.vcf_tokenised_on_device()
        finally:
            del os.environ["EDSX_HOST_TOKENIZER"]
        assert got == host == _want(vcf, fasta), it


def test_large_outputs_every_size_remainder(ctx):
    """.eds of 17 MB and more (pinned-chunk download): reference lengths with every remainder modulo the copy threads
    and the 64-byte rounding, a handful of records, against the oracle."""
    import numpy as np
    rng = np.random.default_rng(3)
    base = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 17_825_800, dtype=np.uint8)].tobytes()
    hdr = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS0\tS1\n"
    for extra in range(0, 12):
        ref = base[:17_825_792 - 40 + extra]          # .eds = reference + a few braces: sizes step through all remainders
        fasta = b">c\n" + ref + b"\n"
        recs = [(1000, "0|1\t1|1"), (9_000_000, "1|0\t0|0"), (len(ref) - 5, "0|0\t0|1")]
        vcf = (hdr + "".join("c\t%d\t.\t%s\t%s\t.\t.\t.\tGT\t%s\n" % (p, chr(ref[p - 1]), "A" if chr(ref[p - 1]) != "A" else "C", g)
                             for p, g in recs)).encode()
        want = o.vcf(vcf, fasta, 0)
        got = ctx.vcf_transform(vcf, fasta, 0)
        assert len(got[0]) >= 16 << 20
        assert got == want, extra


def test_text_length_every_remainder_mod_8_with_and_without_final_newline(ctx):
    """The device tokeniser reads the text through aligned 8-byte words (ByteWindow, csrc/vcf_device.hip): the last line
    may end anywhere inside its word.  Text lengths 0..7 modulo 8 (a comment line of adjustable length in front), with
    and without the final newline, last record with a long ALT, an <INS> (copies REF) and a <DEL>; and a text whose
    length is just below / at / above multiples of 256 (the allocation granule of the text buffer)."""
    rng = random.Random(88)
    base_vcf, fasta = _random_vcf(rng, 2000, 60, 4, 60)
    lines = base_vcf.decode().split("\n")
    head, body = lines[:2], [x for x in lines[2:] if x]
    ref = "".join(fasta.decode().split("\n")[1:])
    last_pos = 1990
    tails = [
        "chr1\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t0|1\t1|1\t0|0\t1|0" % (last_pos, ref[last_pos - 1], ref[last_pos - 1] + "ACGTACGTAC"),
        "chr1\t%d\t.\t%s\t<INS>\t.\tPASS\t.\tGT\t0|1\t1|1\t0|0\t1|0" % (last_pos, ref[last_pos - 1:last_pos + 4]),
        "chr1\t%d\t.\t%s\t<DEL>\t.\tPASS\t.\tGT\t0|1\t1|1\t0|0\t1|0" % (last_pos, ref[last_pos - 1:last_pos + 2]),
    ]
    body = [b for b in body if int(b.split("\t")[1]) < last_pos - 12]
    seen = set()
    for tail in tails:
        for pad in range(0, 8):
            for nl in ("\n", ""):
                vcf = ("\n".join([head[0], "##pad=" + "x" * pad, head[1]] + body + [tail]) + nl).encode()
                seen.add((len(vcf) % 8, nl))
                got = _run(ctx, vcf, fasta, 0)
                assert ctx.vcf_tokenised_on_device()
                assert got == _want(vcf, fasta), (tail[:30], pad, nl)
    assert len(seen) == 16
    for target in (255, 256, 257, 511, 512, 513, 4095, 4096, 4097):
        stem = "\n".join([head[0], head[1]] + body[:3] + [tails[0]])
        padn = target - len(stem) - len("##pad=\n")
        if padn < 0:
            continue
        vcf = ("\n".join([head[0], "##pad=" + "x" * padn, head[1]] + body[:3] + [tails[0]])).encode()
        assert len(vcf) == target
        assert _run(ctx, vcf, fasta, 0) == _want(vcf, fasta), target
