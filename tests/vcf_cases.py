"""Seeded synthetic VCF + FASTA of BASELINE configs[3]'s shape (no oracle imports: also used by profiles/)."""


def gen_vcf(Lf, nrec, ns, seed):
    import random
    rng = random.Random(seed)
    seq = "".join(rng.choices("ACGT", k=Lf))
    fasta = ">chr1 synthetic\n" + "\n".join(seq[i:i + 60] for i in range(0, Lf, 60)) + "\n"
    pos = sorted(rng.sample(range(1, Lf - 12), nrec))
    out = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("s%d" % i for i in range(ns))]
    for p in pos:
        r = rng.random()
        base = seq[p - 1]
        if r < 0.7:
            ref, alt = base, rng.choice([b for b in "ACGT" if b != base])
        elif r < 0.85:
            ref, alt = base, base + "".join(rng.choices("ACGT", k=rng.randint(1, 10)))
        else:
            d = rng.randint(1, 10)
            ref, alt = seq[p - 1:p + d], base
        gts = "\t".join("%d|%d" % (rng.random() < 0.3, rng.random() < 0.3) for _ in range(ns))
        out.append("chr1\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t%s" % (p, ref, alt, gts))
    return ("\n".join(out) + "\n").encode(), fasta.encode()
