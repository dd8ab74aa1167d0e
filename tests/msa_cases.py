"""Seeded random alignments for the MSA parity tests (host-side generator, small sizes)."""
import random


def random_msa(rng, S=None, L=None, lw=None, trailing_newline=True, p_var=0.08, p_gap=0.3):
    S = S or rng.randint(2, 12)
    L = L or rng.randint(1, 200)
    ref = [rng.choice("ACGT") for _ in range(L)]
    # reference-row gaps (insertions in other rows)
    for c in range(L):
        if rng.random() < p_var * 0.3:
            ref[c] = "-"
    rows = [ref]
    var_cols = [c for c in range(L) if rng.random() < p_var]
    for _ in range(S - 1):
        row = list(ref)
        for c in range(L):
            if ref[c] == "-" and rng.random() < 0.5:
                row[c] = rng.choice("ACGT")
        for c in var_cols:
            if rng.random() < 0.5:
                row[c] = "-" if rng.random() < p_gap else rng.choice("ACGTacgtN")
        rows.append(row)
    if lw is None:
        lw = rng.choice([L, L, 7, 60, 3])
    out = []
    for i, row in enumerate(rows):
        out.append(">s%d some text %s" % (i, "x" * rng.randint(0, 9)))
        s = "".join(row)
        for k in range(0, L, lw):
            out.append(s[k:k + lw])
    text = "\n".join(out)
    if trailing_newline:
        text += "\n"
    return text.encode()


CAMPAIGN_ALPHABETS = ["ACGT", "ACGTN", "ACGTacgtN", "ACDEFGHIKLMNPQRSTVWY", "AC"]


def campaign_msa(rng):
    """One alignment of the randomized parity campaign: row counts around every kernel switch (rows per
    lane, S <= 256 instantiations, the 1024 fast/generic limit), column counts around tile edges, DNA /
    lower-case / protein alphabets, site and gap densities from 0 to 1, wrapped lines, header styles,
    missing or doubled final newline.  Returns (bytes, description)."""
    S = rng.choice([2, 3, 5, 17, 33, 64, 65, 100, 130, 256, 257, 400, 513, 700, 960, 1000, 1024, 1025, 1100,
                    rng.randint(2, 1200)])
    L = rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 300, 1000, 2047, 2048, 2049, rng.randint(1, 6000)])
    if S * L > 3_000_000:
        L = max(1, 3_000_000 // S)
    alph = rng.choice(CAMPAIGN_ALPHABETS)
    p_var = rng.choice([0.0, 0.01, 0.05, 0.2, 0.6, 1.0])
    p_gap = rng.choice([0.0, 0.1, 0.5])
    p_row = rng.choice([0.02, 0.3, 0.5, 0.9])
    ref = [rng.choice(alph) for _ in range(L)]
    for c in range(L):
        if rng.random() < p_var * 0.2:
            ref[c] = "-"
    cols = [c for c in range(L) if rng.random() < p_var]
    rows = [ref]
    for _ in range(S - 1):
        row = list(ref)
        for c in cols:
            if rng.random() < p_row:
                row[c] = "-" if rng.random() < p_gap else rng.choice(alph)
        rows.append(row)
    lw = rng.choice([L, L, L, 60, 7, 1, max(1, L // 3)])
    hdr = rng.choice(["s%d", "seq_%06d", "x%d some description", "%d"])
    out = []
    for i, row in enumerate(rows):
        out.append(">" + hdr % i)
        s = "".join(row)
        out.extend(s[k:k + lw] for k in range(0, L, lw))
    trailing = rng.choice(["\n", "\n", "", "\n\n"])
    return ("\n".join(out) + trailing).encode(), (S, L, lw, alph, p_var, p_gap, p_row)


def wide_msa(rng, S=None, nul=False, alph=None):
    """Alignments whose variant segments are WIDE (2..70 adjacent variant columns) and whose rows are drawn
    from a few haplotypes, some of them the same letters with the gaps placed differently (so that different
    raw rows spell one string, also strings of more than 12 letters) — the multi-column signature path of the
    wave-per-segment kernels and its byte-for-byte verification.  nul=True sprinkles NUL bytes into variant
    columns (the reference ends a row's string at '\\0', msa_transforms.cpp:282) and makes some columns NUL
    in every row."""
    S = S or rng.choice([2, 3, 7, 40, 64, 65, 200, 256, 257, 700, 1000, 1024])
    alph = alph or rng.choice(["ACGT", "ACGTN", "ACGTacgtN", "ACDEFGHIKLMNPQRSTVWY"])
    nblocks = rng.randint(1, 6)
    cols = []          # list of per-column lists (one char per row)
    def common(n):
        for _ in range(n):
            ch = rng.choice(alph)
            cols.append([ch] * S)
    common(rng.randint(0, 5))
    for _ in range(nblocks):
        w = rng.choice([2, 5, 10, 11, 12, 13, 20, 21, 30, 63, 64, 65, 70, rng.randint(2, 70)])
        nh = rng.randint(1, 6)
        haps = []
        for h in range(nh):
            letters = [rng.choice(alph) for _ in range(rng.randint(0, w))]
            for variant in range(rng.randint(1, 3)):     # same letters, gaps placed differently
                pos = sorted(rng.sample(range(w), len(letters)))
                row = ["-"] * w
                for p, ch in zip(pos, letters):
                    row[p] = ch
                haps.append(row)
        pick = [rng.randrange(len(haps)) for _ in range(S)]
        # make sure the block is variant in every column w.r.t. row 0 or row 0 has a gap there: force it by
        # giving one row the complement letter where needed
        for c in range(w):
            col = [haps[pick[r]][c] for r in range(S)]
            if col[0] != "-" and all(x == col[0] for x in col):
                r = rng.randrange(1, S) if S > 1 else 0
                col[r] = "-" if rng.random() < 0.5 else rng.choice([a for a in alph if a != col[0]] or ["-"])
            if nul and rng.random() < 0.15:
                for _ in range(rng.randint(1, 3)):
                    col[rng.randrange(S)] = "\0"
            cols.append(col)
        common(rng.randint(1, 40))
        if nul and rng.random() < 0.3:
            cols.append(["\0"] * S)                       # a NUL in every row: a common column
            common(rng.randint(1, 3))
    L = len(cols)
    lw = rng.choice([L, L, 60, 7])
    out = []
    for r in range(S):
        out.append(">w%d" % r)
        s = "".join(cols[c][r] for c in range(L))
        out.extend(s[k:k + lw] for k in range(0, L, lw))
    return ("\n".join(out) + rng.choice(["\n", ""])).encode("latin-1")
