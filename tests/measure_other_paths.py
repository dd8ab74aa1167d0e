#!/usr/bin/env python3
"""Wall-clock of the EDS -> l-EDS merge and the VCF -> EDS overlay through the C ABI (host bytes in, host
bytes out: host tokeniser + device rounds + materialisation) beside the CPU oracle on the same host,
with byte-equality of the two outputs checked.  Inputs: genrandomeds-shaped EDS (+ sEDS) and a
BASELINE-C4-shaped VCF at reduced scale.  Usage: python3 tests/measure_other_paths.py [eds_mb] [vcf_k]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # the oracle is test infrastructure: this script lives under tests/
import torch, edsparser_amd
import oracle_lib as o

def gen_eds(ref_mb, v, seed):
    rng = random.Random(seed)
    L = int(ref_mb * 1e6)
    ref = rng.choices("ACGT", k=L)
    sites = sorted(rng.sample(range(L), int(L * v)))
    eds, seds, cur = [], [], 0
    for p in sites:
        if p > cur:
            eds.append("{" + "".join(ref[cur:p]) + "}"); seds.append("{0}")
        k = rng.randint(2, 4)
        alts = [ref[p]]
        for _ in range(k - 1):
            r = rng.random()
            if r < 0.7: alts.append(rng.choice([b for b in "ACGT" if b != ref[p]]))
            elif r < 0.85: alts.append(ref[p] + "".join(rng.choices("ACGT", k=rng.randint(1, 10))))
            else: alts.append("")
        paths = [pp if pp < k else rng.randrange(k) for pp in range(4)]
        eds.append("{" + ",".join(alts) + "}")
        for a in range(k):
            ids = [str(i + 1) for i, c in enumerate(paths) if c == a]
            seds.append("{" + ",".join(ids) + "}")
        cur = p + 1
    if cur < L:
        eds.append("{" + "".join(ref[cur:]) + "}"); seds.append("{0}")
    return "".join(eds).encode(), "".join(seds).encode()

def gen_vcf(Lf, nrec, ns, seed):
    rng = random.Random(seed)
    seq = "".join(rng.choices("ACGT", k=Lf))
    fasta = ">chr1 synthetic\n" + "\n".join(seq[i:i + 60] for i in range(0, Lf, 60)) + "\n"
    pos = sorted(rng.sample(range(1, Lf - 12), nrec))
    out = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("s%d" % i for i in range(ns))]
    for p in pos:
        r = rng.random()
        base = seq[p - 1]
        if r < 0.7: ref, alt = base, rng.choice([b for b in "ACGT" if b != base])
        elif r < 0.85: ref, alt = base, base + "".join(rng.choices("ACGT", k=rng.randint(1, 10)))
        else:
            d = rng.randint(1, 10); ref, alt = seq[p - 1:p + d], base
        gts = "\t".join("%d|%d" % (rng.random() < 0.3, rng.random() < 0.3) for _ in range(ns))
        out.append("chr1\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t%s" % (p, ref, alt, gts))
    return ("\n".join(out) + "\n").encode(), fasta.encode()

def timed(f, reps=1):
    best, res = 1e30, None
    for _ in range(reps):
        t0 = time.perf_counter(); res = f(); best = min(best, time.perf_counter() - t0)
    return best, res

def main():
    eds_mb = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    vcf_k = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    ctx = edsparser_amd.Context(0)
    ctx.leds_merge(b"{A}{C,G}{T}", None, 10)                                        # warm-up
    # CARTESIAN output grows with the product of the chained sites (42-95 MB out for 0.1-1 MB in at this
    # shape), so that mode is timed on 1 MB; LINEAR keeps one string per surviving path and is timed on eds_mb
    for name, mb, with_src, l in (("CARTESIAN l=10 (C1 shape)", min(eds_mb, 1.0), False, 10), ("LINEAR l=32 (C3 shape)", eds_mb, True, 32)):
        t0 = time.perf_counter(); eds, seds = gen_eds(mb, 0.05, 42)
        sd = seds if with_src else None
        print("EDS %.1f MB + sEDS %.1f MB generated in %.1f s" % (len(eds) / 1e6, len(seds) / 1e6, time.perf_counter() - t0), flush=True)
        tg, rg = timed(lambda: ctx.leds_merge(eds, sd, l), 2)
        tc, rc = timed(lambda: o.merge(eds, sd, l))
        nin = len(eds) + (len(sd) if sd else 0)
        print("merge %-26s in %.1f MB, out %.1f MB: GPU path %.3f s (%.1f MB/s in) | CPU oracle, 1 core %.3f s (%.1f MB/s) | equal=%s"
              % (name, nin / 1e6, (len(rg[0]) + len(rg[1])) / 1e6, tg, nin / tg / 1e6, tc, nin / tc / 1e6, tuple(rg) == tuple(rc)), flush=True)
    t0 = time.perf_counter(); vcf, fasta = gen_vcf(vcf_k * 100_000, vcf_k * 1000, 8, 7)
    print("VCF %.1f MB (%d records, 8 samples) + FASTA %.1f MB generated in %.1f s" % (len(vcf) / 1e6, vcf_k * 1000, len(fasta) / 1e6, time.perf_counter() - t0), flush=True)
    tg, rg = timed(lambda: ctx.vcf_transform(vcf, fasta, 0), 2)
    tc, rc = timed(lambda: o.vcf(vcf, fasta, 0))
    nin = len(vcf) + len(fasta)
    print("vcf2eds in %.1f MB: GPU path %.3f s (%.1f MB/s) | CPU oracle, 1 core %.3f s (%.1f MB/s) | equal=%s"
          % (nin / 1e6, tg, nin / tg / 1e6, tc, nin / tc / 1e6, tuple(rg[:2]) == tuple(rc[:2])), flush=True)

main()
