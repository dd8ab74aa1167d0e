"""Randomized parity campaign against the oracle (run on the GPU box: python tests/campaign_vcf.py <seed> <cases>).
The suite runs a fixed slice of the same generators (test_randomized_campaign); logs of long runs are in profiles/."""
import os, sys, random, time, resource
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import edsparser_amd, oracle_lib as o
import test_vcf_gpu as T
ctx = edsparser_amd.Context(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = random.Random(seed)
bad = 0; t0 = time.time(); errs = 0
for it in range(ncases):
    L = rng.choice([30, 100, 1000, 5000, 30000])
    dens = rng.choice([0.01, 0.05, 0.2, 0.4])
    nvar = max(1, min(L - 12, int(L * dens)))
    ns = rng.choice([0, 1, 2, 8, 63, 64, 65, 70])
    lw = rng.choice([60, 70, 7, L, 61])
    l = rng.choice([0, 0, 0, 3, 10]) if nvar <= 60 else 0
    vcf, fasta = T._random_vcf(rng, L, nvar, ns, lw)
    if rng.random() < 0.3:                                   # shuffle record order (the tool sorts)
        lines = vcf.decode().split("\n"); head, body = lines[:2], [x for x in lines[2:] if x]
        rng.shuffle(body); vcf = ("\n".join(head + body) + "\n").encode()
    try:
        e, s, st = o.vcf(vcf, fasta, l); want = (e, s, st)
    except o.OracleError as ex:
        want = ("ERR", str(ex)); errs += 1
    try:
        got = ctx.vcf_transform(vcf, fasta, l)
    except edsparser_amd.EdsxError as ex:
        got = ("ERR", ex.message)
    if got != want:
        bad += 1
        print("MISMATCH case", it, (L, nvar, ns, lw, l), str(want)[:160], "|||", str(got)[:160], flush=True)
        open("gpurun_out/vcf_fail_%d_%d.vcf" % (seed, it), "wb").write(vcf); open("gpurun_out/vcf_fail_%d_%d.fa" % (seed, it), "wb").write(fasta)
        if bad >= 3: break
print("vcf campaign seed", seed, "cases", it + 1, "mismatches", bad, "oracle errors", errs, "in %.1f s" % (time.time() - t0), "maxrss MB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024)
