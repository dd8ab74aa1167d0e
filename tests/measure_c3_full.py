"""BASELINE configs[2] at full size on one GPU: eds2leds LINEAR merge with .seds sources, genrandomeds-shaped 100 Mb
reference at the tool's default 10 % sites, l = 32.  Times the single C ABI call, checks it against
the CPU oracle (one core) and against the 8-way symbol-range partition (ranks as threads on the one GPU).
Usage: python tests/measure_c3_full.py [ref_mb]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edsparser_amd  # noqa: E402
import oracle_lib as o  # noqa: E402
from edsparser_amd import multigpu as mg  # noqa: E402
from measure_sharded_paths import run_ranks  # noqa: E402
from merge_cases import genrandomeds_shaped  # noqa: E402


def main():
    mb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
    t0 = time.perf_counter()
    eds, seds = genrandomeds_shaped(mb, 0.10, 42)
    print("generated .eds %.1f MB + .seds %.1f MB in %.0f s" % (len(eds) / 1e6, len(seds) / 1e6, time.perf_counter() - t0), flush=True)
    ctx = edsparser_amd.Context(0)
    ctx.leds_merge(b"{A}{C,G}{T}", None, 10)                       # warm-up
    best, res = 1e9, None
    for _ in range(3):
        t0 = time.perf_counter()
        res = ctx.leds_merge(eds, seds, 32, True)
        best = min(best, time.perf_counter() - t0)
    nin = len(eds) + len(seds)
    print("eds2leds LINEAR l=32 single call (Python wrapper, incl. one extra copy of the outputs): %.3f s = %.1f MB/s of input; out %.1f MB; "
          "tokenised on device: %s" % (best, nin / best / 1e6, (len(res[0]) + len(res[1])) / 1e6, ctx.leds_tokenised_on_device()), flush=True)
    parts, t = run_ranks(8, lambda c, r, w, d: mg.gpu_merge_sharder(c, r, w, d).run(eds, seds, 32, True))
    same = (b"".join(x["leds"] for x in parts), b"".join(x["seds"] for x in parts)) == res
    print("8 symbol ranges (threads on one GPU): %.3f s, partitioned=%s ranges=%d, equal to the single call: %s"
          % (t, parts[0]["partitioned"], parts[0]["ranges"], same), flush=True)
    del parts
    t0 = time.perf_counter()
    want = o.merge(eds, seds, 32, True)
    tc = time.perf_counter() - t0
    print("CPU oracle, 1 core: %.1f s = %.1f MB/s; equal to the GPU output: %s" % (tc, nin / tc / 1e6, want == res), flush=True)


main()
