"""Seeded random inputs for the EDS -> l-EDS merge parity tests."""
import random


def campaign_eds(rng):
    """One input of the randomized merge campaign: LINEAR (path partitions, universal {0} sets, arbitrary
    subsets that may intersect to nothing) or CARTESIAN, 1..3000 symbols, empty and duplicate strings,
    1..130 paths (1..3 bitset words), FULL or COMPACT input text.  The last field of the description is a
    bound on the strings of one merged symbol (product of the set sizes inside a merging chain): callers
    skip cases above 1e5, the reference and the oracle would exhaust memory on them as well."""
    linear = rng.random() < 0.6
    n = rng.choice([1, 2, 3, 10, 50, 300, rng.randint(1, 3000 if linear else 250)])
    p_deg = rng.choice([0.0, 0.1, 0.3, 0.6, 0.95]) if linear else rng.choice([0.0, 0.1, 0.25])
    paths = rng.choice([1, 2, 3, 8, 63, 64, 65, 130])
    case_mode = rng.choice([0.1, 0.1, 0.1, 0.8, 0.9])          # partition (bounded by the paths) / universal / arbitrary
    if not linear or case_mode > 0.7: n = min(n, 24)                # products can explode: keep the chain short
    maxlen = rng.choice([0, 1, 3, 12, 40])
    syms, srcs = [], []
    for _ in range(n):
        k = 1 if rng.random() >= p_deg else rng.randint(2, 4 if linear else 3)
        strs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(0, maxlen))) for _ in range(k)]
        if k > 1 and rng.random() < 0.2: strs[1] = strs[0]            # duplicates
        syms.append(strs)
        mode = case_mode if k > 1 else rng.random()
        if k == 1:
            srcs.append([0] if mode < 0.9 else [rng.randint(1, paths)])
        elif mode < 0.7:                                             # a partition of the paths
            choice = [rng.randrange(k) for _ in range(paths)]
            for a in range(k):
                ids = [p + 1 for p in range(paths) if choice[p] == a]
                srcs.append(ids if ids else [rng.randint(1, paths)])
        elif mode < 0.85:                                            # universal sets mixed in
            for a in range(k): srcs.append([0] if rng.random() < 0.5 else sorted(rng.sample(range(1, paths + 1), rng.randint(1, min(paths, 3)))))
        else:                                                        # arbitrary subsets (may empty out)
            for a in range(k): srcs.append(sorted(rng.sample(range(1, paths + 1), rng.randint(1, min(paths, 4)))))
    l = rng.choice([1, 2, 3, 8, 20, 50]) if linear else rng.choice([1, 2, 3, 4])
    # bound on the strings of one merged symbol: product of the set sizes inside every chain that merges
    prod, cur = 1, 1
    for st in syms:
        if len(st) == 1 and len(st[0]) >= l: cur = 1
        else:
            cur *= len(st); prod = max(prod, cur)
    compact_in = rng.random() < 0.3
    text = "".join((s[0] if (compact_in and len(s) == 1 and s[0]) else "{" + ",".join(s) + "}") for s in syms)
    if rng.random() < 0.2: text = text.replace("}{", "}\n{", 3)
    seds = "".join("{" + ",".join(map(str, s)) + "}" for s in srcs) if linear else None
    return text.encode(), (seds.encode() if seds else None), l, rng.random() < 0.5, (n, p_deg, paths, maxlen, linear, case_mode, prod)


def genrandomeds_shaped(ref_mb, v, seed, paths=4):
    """genrandomeds-shaped .eds / .seds pair: uniform ACGT reference, floor(bp * v) single-base
    sites with 2..4 alternatives (reference base, SNPs, insertions of 1..10 bases, deletions), 4 paths."""
    rng = random.Random(seed)
    L = int(ref_mb * 1e6)
    ref = rng.choices("ACGT", k=L)
    sites = sorted(rng.sample(range(L), int(L * v)))
    eds, seds, cur = [], [], 0
    for p in sites:
        if p > cur:
            eds.append("{" + "".join(ref[cur:p]) + "}")
            seds.append("{0}")
        k = rng.randint(2, 4)
        alts = [ref[p]]
        for _ in range(k - 1):
            r = rng.random()
            if r < 0.7:
                alts.append(rng.choice([b for b in "ACGT" if b != ref[p]]))
            elif r < 0.85:
                alts.append(ref[p] + "".join(rng.choices("ACGT", k=rng.randint(1, 10))))
            else:
                alts.append("")
        choice = [pp if pp < k else rng.randrange(k) for pp in range(paths)]
        eds.append("{" + ",".join(alts) + "}")
        for a in range(k):
            seds.append("{" + ",".join(str(i + 1) for i, c in enumerate(choice) if c == a) + "}")
        cur = p + 1
    if cur < L:
        eds.append("{" + "".join(ref[cur:]) + "}")
        seds.append("{0}")
    return "".join(eds).encode(), "".join(seds).encode()
