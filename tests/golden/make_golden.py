#!/usr/bin/env python3
"""Generate golden vectors for the merge and VCF paths by running the REAL reference library
(oracle/_ref/libedsref.so, compiled from /root/reference by oracle/Makefile) on seeded random
inputs.  Runs only in the build container (the reference does not travel); its outputs —
tests/golden/gen_merge.json and tests/golden/gen_vcf.json — are committed as data fixtures.

    python tests/golden/make_golden.py            # regenerate fixtures
    python tests/golden/make_golden.py --check    # re-run reference, compare with committed fixtures
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as o  # noqa: E402


def rand_str(rng, lo, hi, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(rng.randint(lo, hi)))


def gen_merge_case(rng):
    n = rng.randint(1, 9)
    paths = rng.randint(1, 5)
    with_src = rng.random() < 0.55
    syms, srcs = [], []
    for _ in range(n):
        k = 1 if rng.random() < 0.5 else rng.randint(2, 4)
        strs = [rand_str(rng, 0, 6) for _ in range(k)]
        syms.append(strs)
        for _s in strs:
            if k == 1 and rng.random() < 0.7:
                srcs.append([0])
            else:
                ids = sorted(set(rng.randint(0 if rng.random() < 0.15 else 1, paths)
                                 for _ in range(rng.randint(1, 3))))
                srcs.append(ids)
    compact_in = rng.random() < 0.3
    text = ""
    for strs in syms:
        if compact_in and len(strs) == 1 and strs[0]:
            text += strs[0]
        else:
            text += "{" + ",".join(strs) + "}"
    if rng.random() < 0.2:
        text += "\n"
    seds = "".join("{" + ",".join(map(str, s)) + "}" for s in srcs) if with_src else None
    return {"eds": text, "seds": seds, "l": rng.randint(1, 6), "compact": rng.random() < 0.7}


def run_merge(fn, c):
    try:
        out, so = fn(c["eds"].encode(), c["seds"].encode() if c["seds"] is not None else None,
                     c["l"], c["compact"])
        return {"out": out.decode(), "seds_out": so.decode()}
    except o.OracleError as ex:
        return {"error": str(ex)}


def gen_vcf_case(rng):
    L = rng.randint(12, 80)
    ref = rand_str(rng, L, L)
    lw = rng.choice([L, 10, 7, 60])
    fasta = ">chr1 test\n" + "\n".join(ref[i:i + lw] for i in range(0, L, lw)) + "\n"
    if rng.random() < 0.2:
        fasta += ">chr2\nACGT\n"
    ns = rng.choice([0, 1, 2, 4])
    recs = []
    positions = rng.sample(range(1, L + 1), min(L, rng.randint(1, 8)))
    for p in positions:
        reflen = 1 if rng.random() < 0.7 else rng.randint(1, 4)
        r = ref[p - 1:p - 1 + reflen] or "A"
        if rng.random() < 0.1:
            r = rand_str(rng, len(r), len(r))     # VCF REF differing from FASTA
        nalt = 1 if rng.random() < 0.7 else rng.randint(2, 3)
        alts = []
        for _ in range(nalt):
            x = rng.random()
            if x < 0.55:
                alts.append(rand_str(rng, 1, 1))
            elif x < 0.75:
                alts.append(r[0] + rand_str(rng, 1, 4))
            elif x < 0.85:
                alts.append("<DEL>")
            elif x < 0.9:
                alts.append("<INS>")
            elif x < 0.93:
                alts.append("<INV>")
            else:
                alts.append(rand_str(rng, 1, 3))
        gts = []
        for _s in range(ns):
            x = rng.random()
            sep = "|" if rng.random() < 0.8 else "/"
            if x < 0.08:
                gt = "." + sep + "."
            elif x < 0.15:
                gt = str(rng.randint(0, nalt))
            else:
                gt = sep.join(str(rng.randint(0, nalt)) for _ in range(2))
            if rng.random() < 0.2:
                gt += ":12"
            gts.append(gt)
        fields = ["chr1", str(p), ".", r, ",".join(alts), "99", "PASS", "."]
        if ns:
            fields += ["GT"] + gts
        recs.append(fields)
    if rng.random() < 0.6:
        recs.sort(key=lambda f: int(f[1]))
    sep = "\t" if rng.random() < 0.85 else " "
    lines = ["##fileformat=VCFv4.2"]
    hdr = ["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO"]
    if ns:
        hdr += ["FORMAT"] + ["S%d" % i for i in range(ns)]
    lines.append("\t".join(hdr))
    for f in recs:
        lines.append(sep.join(f))
        if rng.random() < 0.05:
            lines.append("badline")
    vcf = "\n".join(lines) + "\n"
    return {"vcf": vcf, "fasta": fasta, "l": 0 if rng.random() < 0.7 else rng.randint(1, 5)}


def run_vcf(fn, c):
    try:
        e, s, st = fn(c["vcf"].encode(), c["fasta"].encode(), c["l"])
        return {"eds": e.decode(), "seds": s.decode(), "stats": st}
    except o.OracleError as ex:
        return {"error": str(ex)}


def main():
    check = "--check" in sys.argv
    if not o.have_ref():
        sys.exit("oracle/_ref/libedsref.so not built (needs /root/reference): run make -C oracle")
    rng = random.Random(20251114)
    merge_cases = []
    for _ in range(400):
        c = gen_merge_case(rng)
        c["expect"] = run_merge(o.ref_merge, c)
        merge_cases.append(c)
    rng = random.Random(777)
    vcf_cases = []
    while len(vcf_cases) < 300:
        c = gen_vcf_case(rng)
        # equal-POS records permute under std::sort only for >16 records; cases here have <= 8
        c["expect"] = run_vcf(o.ref_vcf, c)
        vcf_cases.append(c)
    prov = ("Generated by tests/golden/make_golden.py from the reference library compiled in the "
            "build container (oracle/_ref); inputs are seeded random, 'expect' is the reference's output.")
    for name, cases in (("gen_merge.json", merge_cases), ("gen_vcf.json", vcf_cases)):
        path = os.path.join(HERE, name)
        doc = {"_provenance": prov, "cases": cases}
        if check:
            old = json.load(open(path))
            assert old["cases"] == cases, name + " differs from the reference's current output"
            print("check ok:", name, len(cases))
        else:
            json.dump(doc, open(path, "w"), indent=0)
            print("wrote", path, len(cases))
    # oracle restatement vs reference on the same cases
    bad = 0
    for c in merge_cases:
        if run_merge(o.merge, c) != c["expect"]:
            bad += 1
            print("MERGE MISMATCH", c, run_merge(o.merge, c))
    for c in vcf_cases:
        if run_vcf(o.vcf, c) != c["expect"]:
            bad += 1
            print("VCF MISMATCH", c, run_vcf(o.vcf, c))
    print("oracle-vs-reference mismatches:", bad)
    n_err = sum("error" in c["expect"] for c in merge_cases), sum("error" in c["expect"] for c in vcf_cases)
    print("reference error cases (merge, vcf):", n_err)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
