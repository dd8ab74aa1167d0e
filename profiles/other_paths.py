#!/usr/bin/env python3
"""Device-side view of the merge and VCF paths for rocprofv3 --kernel-trace --stats (profiles/collect_other.sh):
BASELINE configs[2] shape (LINEAR, l = 32, genrandomeds 10 % sites) and configs[3] shape at 1/10 scale, three calls
each through the host-buffer C ABI.  Prints input / output bytes so the kernel times can be set against the
algorithmic bytes of SURVEY §8(d).  No oracle here (parity of these shapes: tests/test_*_shard_gpu.py)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import edsparser_amd  # noqa: E402
from merge_cases import genrandomeds_shaped  # noqa: E402
from vcf_cases import gen_vcf  # noqa: E402

ctx = edsparser_amd.Context(0)
eds, seds = genrandomeds_shaped(10, 0.10, 42)
vcf, fasta = gen_vcf(100_000_000, 1_000_000, 8, 7)
for _ in range(3):
    t0 = time.perf_counter()
    out, sout = ctx.leds_merge(eds, seds, 32, True)
    t1 = time.perf_counter()
    e, s, st = ctx.vcf_transform(vcf, fasta, 0)
    t2 = time.perf_counter()
    print("merge: in %d B out %d B %.3f s | vcf: in %d B out %d B %.3f s" %
          (len(eds) + len(seds), len(out) + len(sout), t1 - t0, len(vcf) + len(fasta), len(e) + len(s), t2 - t1), flush=True)
