"""BASELINE configs[3] at full size on one GPU: 1 Gb reference (60-column FASTA) + 10^7-record synthetic VCF (sorted,
distinct POS, 70 % SNP / 15 % insertion 1-10 / 15 % deletion 1-10, 8 diploid phased samples — SURVEY §8(d)).
Times the single C ABI call, checks it against the CPU oracle (one core, timed as the CPU side) and against the 8-way
position-range partition (ranks as threads on the one GPU).  Usage: python tests/measure_c4_full.py [scale] [shuffle]  (1.0 = full)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edsparser_amd  # noqa: E402
import oracle_lib as o  # noqa: E402
from edsparser_amd import multigpu as mg  # noqa: E402
from measure_sharded_paths import run_ranks  # noqa: E402


def gen(Lf, nrec, ns, seed):
    rng = np.random.default_rng(seed)
    Lf -= Lf % 60
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, Lf, dtype=np.uint8)]
    fasta = b">chr1 synthetic\n" + np.concatenate((bases.reshape(-1, 60), np.full((Lf // 60, 1), 10, dtype=np.uint8)), axis=1).tobytes()
    seq = bases.tobytes()
    pos = np.unique(rng.integers(1, Lf - 12, int(nrec * 1.05)))
    pos = np.sort(rng.choice(pos, nrec, replace=False))
    kind = rng.random(nrec)
    ins = ["".join("ACGT"[c] for c in rng.integers(0, 4, rng.integers(1, 11))) for _ in range(4096)]
    gts = ["\t".join("%d|%d" % (a, b) for a, b in (rng.random((ns, 2)) < 0.3).astype(int)) for _ in range(4096)]
    pick = rng.integers(0, 4096, nrec)
    dl = rng.integers(1, 11, nrec)
    other = {"A": "CGT", "C": "AGT", "G": "ACT", "T": "ACG"}
    alt3 = rng.integers(0, 3, nrec)
    out = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("s%d" % i for i in range(ns))]
    for p, k, g, d, a in zip(pos.tolist(), kind.tolist(), pick.tolist(), dl.tolist(), alt3.tolist()):
        base = chr(seq[p - 1])
        if k < 0.7:
            out.append("chr1\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t%s" % (p, base, other[base][a], gts[g]))
        elif k < 0.85:
            out.append("chr1\t%d\t.\t%s\t%s%s\t.\tPASS\t.\tGT\t%s" % (p, base, base, ins[g], gts[g]))
        else:
            out.append("chr1\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t%s" % (p, seq[p - 1:p + d].decode(), base, gts[g]))
    return ("\n".join(out) + "\n").encode(), fasta


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    t0 = time.perf_counter()
    vcf, fasta = gen(int(1_000_000_000 * scale), int(10_000_000 * scale), 8, 7)
    if len(sys.argv) > 2 and sys.argv[2] == "shuffle":            # record lines in random order (distinct POS: device radix sort)
        import random
        lines = vcf.split(b"\n")
        head, body = lines[:2], [x for x in lines[2:] if x]
        random.Random(5).shuffle(body)
        vcf = b"\n".join(head + body) + b"\n"
    print("generated VCF %.1f MB + FASTA %.1f MB in %.0f s" % (len(vcf) / 1e6, len(fasta) / 1e6, time.perf_counter() - t0), flush=True)
    os.environ["EDSX_TRACE"] = "1"
    ctx = edsparser_amd.Context(0)
    ctx.vcf_transform(vcf[:200000].rsplit(b"\n", 1)[0] + b"\n", fasta[:2_000_000], 0)
    best, res = 1e9, None
    for _ in range(3):
        t0 = time.perf_counter()
        res = ctx.vcf_transform(vcf, fasta, 0)
        best = min(best, time.perf_counter() - t0)
    nin = len(vcf) + len(fasta)
    print("vcf2eds single call (Python wrapper, incl. one extra copy of the outputs): %.3f s = %.1f MB/s of input; out %.1f MB; on device: %s; stats %s"
          % (best, nin / best / 1e6, (len(res[0]) + len(res[1])) / 1e6, ctx.vcf_tokenised_on_device(), res[2]), flush=True)
    del os.environ["EDSX_TRACE"]
    parts, t = run_ranks(8, lambda c, r, w, d: mg.gpu_vcf_sharder(c, r, w, d).run(vcf, fasta))
    same = (b"".join(x["eds"] for x in parts), b"".join(x["seds"] for x in parts), parts[0]["stats"]) == res
    print("8 position ranges (threads on one GPU): %.3f s, records per range %s, lines moved %d B, equal to the single call: %s"
          % (t, [x["records"] for x in parts], sum(x["moved_lines_bytes"] for x in parts), same), flush=True)
    del parts
    t0 = time.perf_counter()
    want = o.vcf(vcf, fasta, 0)
    tc = time.perf_counter() - t0
    print("CPU oracle, 1 core: %.1f s = %.1f MB/s; equal to the GPU output: %s" % (tc, nin / tc / 1e6, want == res), flush=True)


main()
