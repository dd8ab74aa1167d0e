#!/usr/bin/env python3
"""Second set of reference-generated fixtures (round 2): the inputs the small cases of make_golden.py cannot
reach.  Runs the REAL reference library (oracle/_ref/libedsref.so, compiled in place from /root/reference by
oracle/Makefile) in the build container only; the outputs are committed as data:

  gen2_vcf.json    VCFs of 17..200 records with many equal POS values (libstdc++'s unstable std::sort decides the
                   record order, vcf_transforms.cpp:715-718 / SURVEY quirk 29), and the SURVEY quirk 22-27 shapes
                   (unsorted + overlap + multi-allelic <DEL>, no samples + <INV> + bad line, hom-ALT / missing /
                   haploid GT, l > 0, POS beyond the reference, REF straddling the end).
  gen2_stats.json  EDS::Statistics + is_leds(l) (eds.cpp:361-505, eds_transforms.cpp:439-468) of the inputs AND of the
                   reference's outputs of every case in gen_merge.json and of the chain cases below (cases refer to
                   their texts by index, the numbers are the reference's).
  gen2_merge.json  merges of 500..2000 symbols that take 6 and more rounds (LINEAR with 1 and 3 threads, and
                   CARTESIAN), plus BASELINE configs[0]'s shape (genrandomeds @5 %, eds2leds -l 10, CARTESIAN) at
                   0.2 MB.  Expected outputs above 64 KB are stored as length + SHA-256 of the reference's bytes.

    python tests/golden/make_golden2.py            # regenerate (minutes: the reference merge is quadratic)
    python tests/golden/make_golden2.py --check    # re-run the reference, compare with the committed fixtures
"""
import hashlib
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as o  # noqa: E402
from merge_cases import genrandomeds_shaped  # noqa: E402

BIG = 64 * 1024


def pack(b):
    """Expected bytes: verbatim when small, else length + sha256."""
    if len(b) <= BIG:
        return {"text": b.decode()}
    return {"len": len(b), "sha256": hashlib.sha256(b).hexdigest()}


def matches(packed, b):
    if "text" in packed:
        return packed["text"].encode() == b
    return packed["len"] == len(b) and packed["sha256"] == hashlib.sha256(b).hexdigest()


# ---------------------------------------------------------------------------------------------- VCF
def samepos_vcf(rng):
    L = rng.randint(60, 400)
    ref = "".join(rng.choice("ACGT") for _ in range(L))
    lw = rng.choice([L, 60, 17])
    fasta = ">chr1\n" + "\n".join(ref[i:i + lw] for i in range(0, L, lw)) + "\n"
    ns = rng.choice([1, 2, 4])
    nrec = rng.randint(17, 200)
    npos = rng.randint(1, max(1, nrec // rng.choice([2, 4, 8, 20])))
    pool = rng.sample(range(1, L - 4), min(npos, L - 5))
    recs = []
    for _ in range(nrec):
        p = rng.choice(pool)
        reflen = 1 if rng.random() < 0.8 else rng.randint(2, 4)
        r = ref[p - 1:p - 1 + reflen]
        nalt = 1 if rng.random() < 0.75 else 2
        alts = []
        for _a in range(nalt):
            x = rng.random()
            if x < 0.6:
                alts.append(rng.choice("ACGT"))
            elif x < 0.85:
                alts.append(r[0] + "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 3))))
            else:
                alts.append("<DEL>")
        gts = ["|".join(str(rng.randint(0, nalt)) for _ in range(2)) for _s in range(ns)]
        recs.append(["chr1", str(p), ".", r, ",".join(alts), ".", "PASS", ".", "GT"] + gts)
    order = rng.random()
    if order < 0.4:
        recs.sort(key=lambda f: int(f[1]))              # equal POS stay in file order going in
    hdr = ["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"] + ["S%d" % i for i in range(ns)]
    vcf = "##fileformat=VCFv4.2\n" + "\t".join(hdr) + "\n" + "".join("\t".join(f) + "\n" for f in recs)
    return {"vcf": vcf, "fasta": fasta, "l": 0 if rng.random() < 0.8 else rng.randint(1, 4)}


F20 = ">chr1\nACGTACGTAC\nGTACGTACGT\n"
H2 = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2\n"
QUIRKS = [
    ("q22_unsorted_overlap_multiallelic_del", H2 + "chr1\t12\t.\tT\tC\t.\t.\t.\tGT\t0|1\t1|1\n"
     "chr1\t3\t.\tGTA\tG\t.\t.\t.\tGT\t1|0\t0|0\nchr1\t4\t.\tT\tC,<DEL>\t.\t.\t.\tGT\t0|1\t2|0\n", F20, 0),
    ("q23_no_samples_inv_badline_dot_alt", "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
     "chr1\t3\t.\tG\tT\t.\t.\t.\nbadline\nchr1\t5\t.\tA\t<INV>\t.\t.\t.\nchr1\t9\t.\tA\t.\t.\t.\t.\n", F20, 0),
    ("q24_hom_alt_missing_haploid", H2 + "chr1\t1\t.\tA\tT\t.\t.\t.\tGT\t1|1\t1|1\nchr1\t20\t.\tT\tG\t.\t.\t.\tGT\t./.\t1\n", F20, 0),
    ("q25_header_desc_second_record_gtdp_unphased_l4", "##fileformat=VCFv4.2\n##contig=<ID=chr1>\n" + H2.split("\n", 1)[1] +
     "chr1\t5\t.\tA\tG\t.\t.\t.\tGT:DP\t0/1:7\t1/1:9\nchr1\t7\t.\tG\tC\t.\t.\t.\tGT:DP\t0/1:3\t0/0:4\n",
     ">chr1 some description\nACGTACGTAC\nGTACGTACGT\n>chr2\nTTTT\n", 4),
    ("q27_pos_beyond_reference", "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n"
     "chr1\t3\t.\tG\tT\t.\t.\t.\tGT\t0|1\nchr1\t15\t.\tA\tG\t.\t.\t.\tGT\t1|1\n", ">chr1\nACGTACGTAC\n", 0),
    ("q27_ref_straddles_the_end", "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n"
     "chr1\t9\t.\tACGG\tA\t.\t.\t.\tGT\t0|1\n", ">chr1\nACGTACGTAC\n", 0),
]


def run_vcf(fn, c):
    try:
        e, s, st = fn(c["vcf"].encode(), c["fasta"].encode(), c["l"])
        return {"eds": e.decode(), "seds": s.decode(), "stats": st}
    except o.OracleError as ex:
        return {"error": str(ex)}


# -------------------------------------------------------------------------------------------- merge
def chain_eds(rng, nsym, linear):
    """Long chains of short symbols between long common blocks: a chain of m mergeable pairs needs ~log2(m) rounds."""
    paths = rng.choice([3, 4, 8])
    syms, srcs = [], []
    while len(syms) < nsym:
        syms.append(["".join(rng.choice("ACGT") for _ in range(rng.randint(40, 60)))])      # a long common block
        srcs.append([0])
        for _ in range(rng.randint(20, 140)):                                                 # the chain
            if rng.random() < (0.06 if linear else 0.04):
                k = rng.randint(2, 3)
                strs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(0, 2))) for _ in range(k)]
                syms.append(strs)
                if linear:
                    choice = [rng.randrange(k) if p >= k else p for p in range(paths)]
                    for a in range(k):
                        srcs.append([p + 1 for p in range(paths) if choice[p] == a])
                else:
                    srcs += [[0]] * k
            else:
                syms.append(["".join(rng.choice("ACGT") for _ in range(rng.randint(1, 3)))])
                srcs.append([0])
    text = "".join("{" + ",".join(s) + "}" for s in syms)
    seds = "".join("{" + ",".join(map(str, s)) + "}" for s in srcs) if linear else None
    return text, seds


def run_merge(fn, c, **kw):
    try:
        out, so = fn(c["eds"].encode(), c["seds"].encode() if c["seds"] is not None else None, c["l"], c["compact"], **kw)
        return {"out": pack(out), "seds_out": pack(so)}
    except o.OracleError as ex:
        return {"error": str(ex)}


def rounds_of(eds, l):
    """Rounds the merge needs (symbols short of l between degenerate/short neighbours keep pairing up)."""
    import re
    syms = re.findall(r"\{([^}]*)\}", eds)
    runs, cur = [], 0
    for s in syms:
        strs = s.split(",")
        if len(strs) == 1 and len(strs[0]) >= l:
            runs.append(cur); cur = 0
        else:
            cur += 1
    runs.append(cur)
    m, r = max(runs) + 1, 0
    while m > 1:
        m = (m + 1) // 2; r += 1
    return r


def stats_texts(sc, gen_merge_cases, gen2_merge_cases):
    """(eds, seds) bytes a statistics case refers to."""
    c = (gen_merge_cases if sc["src"] == "gen_merge" else gen2_merge_cases)[sc["index"]]
    has_src = c["seds"] is not None
    if sc["what"] == "input":
        return c["eds"].encode(), c["seds"].encode() if has_src else None
    ex = c["expect"]
    out = ex["out"]["text"] if isinstance(ex["out"], dict) else ex["out"]
    so = ex["seds_out"]["text"] if isinstance(ex["seds_out"], dict) else ex["seds_out"]
    return out.encode(), so.encode() if has_src else None


def stats_case_run(fn, sc, gen_merge_cases, gen2_merge_cases):
    eds, seds = stats_texts(sc, gen_merge_cases, gen2_merge_cases)
    try:
        d = fn(eds, seds, sc["l"])
        d = dict(d)
        d.pop("num_context_blocks", None); d.pop("total_paths", None)
        return d
    except Exception as ex:  # noqa: BLE001 — OracleError / EdsxError
        return {"error": getattr(ex, "message", None) or str(ex)}


def main():
    check = "--check" in sys.argv
    if not o.have_ref():
        sys.exit("oracle/_ref/libedsref.so not built (needs /root/reference): run make -C oracle")
    t0 = time.time()
    rng = random.Random(2902)
    vcf_cases = []
    for name, vcf, fasta, l in QUIRKS:
        vcf_cases.append({"name": name, "vcf": vcf, "fasta": fasta, "l": l})
    for i in range(60):
        c = samepos_vcf(rng)
        c["name"] = "samepos_%02d" % i
        vcf_cases.append(c)
    for c in vcf_cases:
        c["expect"] = run_vcf(o.ref_vcf, c)
    print("vcf: %d cases, %.1f s" % (len(vcf_cases), time.time() - t0))

    rng = random.Random(5511)
    merge_cases = []
    for i in range(10):
        linear = i < 7
        text, seds = chain_eds(rng, rng.randint(500, 2000), linear)
        c = {"name": "chain_%02d_%s" % (i, "linear" if linear else "cartesian"), "eds": text, "seds": seds,
             "l": rng.choice([4, 5, 8, 12]), "compact": rng.random() < 0.6}
        c["rounds"] = rounds_of(text, c["l"])
        merge_cases.append(c)
    for c in merge_cases:
        t1 = time.time()
        c["expect"] = run_merge(o.ref_merge, c, threads=1)
        if c["seds"] is not None:
            c["expect_threads3"] = run_merge(o.ref_merge, c, threads=3)
        print("  merge %s: %d bytes, %d rounds, %.1f s" % (c["name"], len(c["eds"]), c["rounds"], time.time() - t1), flush=True)
    # BASELINE configs[0]: genrandomeds 5 % sites, eds2leds -l 10 without -s (CARTESIAN), compact output; 0.2 MB reference.
    # The input is regenerated by the tests from (ref_mb, v, seed): tests/merge_cases.genrandomeds_shaped
    c0 = {"name": "configs0_genrandomeds_5pct_cartesian_l10", "generator": {"ref_mb": 0.2, "v": 0.05, "seed": 42},
          "l": 10, "compact": True}
    eds, _seds = genrandomeds_shaped(0.2, 0.05, 42)
    c0["eds_len"] = len(eds)
    c0["eds_sha256"] = hashlib.sha256(eds).hexdigest()
    t1 = time.time()
    out, so = o.ref_merge(eds, None, 10, True, threads=1)
    c0["expect"] = {"out": pack(out), "seds_out": pack(so)}
    print("  merge %s: %d bytes in, %d out, %.1f s" % (c0["name"], len(eds), len(out), time.time() - t1), flush=True)

    # ---- statistics / is_leds of inputs and outputs, by the reference's EDS class
    def ref_stats(eds, seds, l):
        try:
            d = o.ref_eds_stats(eds, seds, l)
            d.pop("num_context_blocks"); d.pop("total_paths")      # not part of EDS::Statistics
            return d
        except o.OracleError as ex:
            return {"error": str(ex)}
    stats_cases = []
    gm = json.load(open(os.path.join(HERE, "gen_merge.json")))["cases"]
    for src, cases in (("gen_merge", gm), ("gen2_merge", merge_cases)):
        for i, c in enumerate(cases):
            seds = c["seds"].encode() if c["seds"] is not None else None
            stats_cases.append({"src": src, "index": i, "what": "input", "l": c["l"], "stats": ref_stats(c["eds"].encode(), seds, c["l"])})
            ex = c["expect"]
            if "error" not in ex and "text" in ex.get("out", {"text": ex.get("out")}) if isinstance(ex.get("out"), dict) else "error" not in ex:
                out = ex["out"]["text"] if isinstance(ex["out"], dict) else ex["out"]
                so = ex["seds_out"]["text"] if isinstance(ex["seds_out"], dict) else ex["seds_out"]
                stats_cases.append({"src": src, "index": i, "what": "output", "l": c["l"],
                                    "stats": ref_stats(out.encode(), so.encode() if seds is not None else None, c["l"])})
    print("stats: %d cases" % len(stats_cases))

    prov = ("Generated by tests/golden/make_golden2.py from the reference library compiled in the build container "
            "(oracle/_ref); 'expect' is the reference's output (bytes above 64 KB as length + sha256).")
    docs = {"gen2_vcf.json": {"_provenance": prov, "cases": vcf_cases},
            "gen2_merge.json": {"_provenance": prov, "cases": merge_cases, "configs0": c0},
            "gen2_stats.json": {"_provenance": prov, "cases": stats_cases}}
    for name, doc in docs.items():
        path = os.path.join(HERE, name)
        if check:
            old = json.load(open(path))
            assert old == json.loads(json.dumps(doc)), name + " differs from the reference's current output"
            print("check ok:", name)
        else:
            json.dump(doc, open(path, "w"), indent=0)
            print("wrote", path, os.path.getsize(path), "bytes")
    # the oracle restatement against the same vectors
    bad = 0
    for c in vcf_cases:
        if run_vcf(o.vcf, c) != c["expect"]:
            bad += 1; print("VCF MISMATCH", c["name"])
    for c in merge_cases:
        if run_merge(o.merge, c) != c["expect"]:
            bad += 1; print("MERGE MISMATCH", c["name"])
        if "expect_threads3" in c and c["expect_threads3"] != c["expect"]:
            bad += 1; print("REFERENCE threads 1 vs 3 differ", c["name"])
    for sc in stats_cases:
        got = stats_case_run(o.eds_stats, sc, gm, merge_cases)
        if got != sc["stats"]:
            bad += 1; print("STATS MISMATCH", sc["src"], sc["index"], sc["what"], got, sc["stats"])
    out, so = o.merge(eds, None, 10, True)
    if not (matches(c0["expect"]["out"], out) and matches(c0["expect"]["seds_out"], so)):
        bad += 1; print("MERGE MISMATCH configs0")
    print("oracle-vs-reference mismatches:", bad, "(%.0f s)" % (time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
