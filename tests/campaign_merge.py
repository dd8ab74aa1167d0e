"""Randomized parity campaign against the oracle (run on the GPU box: python tests/campaign_merge.py <seed> <cases>).
The suite runs a fixed slice of the same generators (test_randomized_campaign); logs of long runs are in profiles/."""
import os, sys, random, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import edsparser_amd, oracle_lib as o
ctx = edsparser_amd.Context(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = random.Random(seed)
def gen(rng):
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
bad = 0; t0 = time.time(); errs = 0
for it in range(ncases):
    eds, seds, l, compact, desc = gen(rng)
    if desc[-1] > 100000: continue                                  # at most 1e5 product strings
    try:
        w = o.merge(eds, seds, l, compact); want = (w[0], w[1])
    except o.OracleError as ex:
        want = ("ERR", str(ex)); errs += 1
    try:
        g = ctx.leds_merge(eds, seds, l, compact); got = (g[0], g[1])
    except edsparser_amd.EdsxError as ex:
        got = ("ERR", ex.message)
    if got != want:
        bad += 1
        print("MISMATCH case", it, desc, "l", l, compact, str(want)[:150], "|||", str(got)[:150], flush=True)
        open("gpurun_out/merge_fail_%d_%d.eds" % (seed, it), "wb").write(eds)
        if seds: open("gpurun_out/merge_fail_%d_%d.seds" % (seed, it), "wb").write(seds)
        if bad >= 3: break
print("merge campaign seed", seed, "cases", it + 1, "mismatches", bad, "oracle errors", errs, "in %.1f s" % (time.time() - t0))
