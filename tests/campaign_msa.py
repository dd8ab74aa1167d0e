"""Randomized parity campaign against the oracle (run on the GPU box: python tests/campaign_msa.py <seed> <cases> [big]
- `big`: row counts of the generic workgroup-per-segment kernels, 1025 .. 8192; `huge`: 8193 .. 40000 rows, their tables in HBM).
The suite runs a fixed slice of the same generators (test_randomized_campaign); logs of long runs are in profiles/."""
import os, sys, random, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import edsparser_amd, oracle_lib as o
ctx = edsparser_amd.Context(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
big = len(sys.argv) > 3 and sys.argv[3] == "big"
huge = len(sys.argv) > 3 and sys.argv[3] == "huge"
rng = random.Random(seed)
ALPH = ["ACGT", "ACGTN", "ACGTacgtN", "ACDEFGHIKLMNPQRSTVWY", "AC"]
def gen(rng):
    S = rng.choice([2, 3, 5, 17, 33, 64, 65, 100, 130, 256, 257, 400, 513, 700, 960, 1000, 1024, 1025, 1100, rng.randint(2, 1200)])
    if big: S = rng.choice([1025, 1500, 2000, 2047, 2048, 2049, 2500, 3000, 4096, 5000, 8191, 8192, rng.randint(1025, 8192)])
    if huge: S = rng.choice([8193, 9000, 16384, 20000, 40000, rng.randint(8193, 40000)])
    L = rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 300, 1000, 2047, 2048, 2049, rng.randint(1, 6000)])
    if S * L > 3_000_000: L = max(1, 3_000_000 // S)
    alph = rng.choice(ALPH)
    p_var = rng.choice([0.0, 0.01, 0.05, 0.2, 0.6, 1.0]); p_gap = rng.choice([0.0, 0.1, 0.5]); p_row = rng.choice([0.02, 0.3, 0.5, 0.9])
    ref = [rng.choice(alph) for _ in range(L)]
    for c in range(L):
        if rng.random() < p_var * 0.2: ref[c] = "-"
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
bad = 0
t0 = time.time()
for it in range(ncases):
    msa, desc = gen(rng)
    for l in (0, rng.choice([1, 2, 5, 9, 33])):
        try:
            want = o.msa(msa, l)
        except o.OracleError as ex:
            want = ("ERR", str(ex))
        try:
            got = ctx.msa_transform(msa, l)
        except edsparser_amd.EdsxError as ex:
            got = ("ERR", ex.message)
        if got != want:
            bad += 1
            print("MISMATCH case", it, "l", l, desc, str(want)[:120], "|||", str(got)[:120], flush=True)
            open("gpurun_out/msa_fail_%d_%d_%d.msa" % (seed, it, l), "wb").write(msa)
            break
    if bad >= 3: break
print("campaign seed", seed, "cases", it + 1, "mismatches", bad, "in %.1f s" % (time.time() - t0))
