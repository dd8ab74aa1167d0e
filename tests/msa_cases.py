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
