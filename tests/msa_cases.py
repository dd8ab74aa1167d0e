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
