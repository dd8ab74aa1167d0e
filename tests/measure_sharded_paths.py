"""Timing of the partitioned merge / VCF drivers (multigpu.MergeSharder / VcfSharder) with N ranks simulated as
threads sharing the box's one GPU, against the single call.  Not a scaling measurement (one GPU, one CPU share):
it shows what the partition logic itself costs (index pass, all-gathers, cut planning, line routing) and that
the pieces equal the whole at this size.  Usage: python tests/measure_sharded_paths.py [eds_mb] [vcf_k_records]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edsparser_amd  # noqa: E402
from edsparser_amd import multigpu as mg  # noqa: E402
from merge_cases import genrandomeds_shaped  # noqa: E402
from vcf_cases import gen_vcf  # noqa: E402
from test_vcf_shard_cpu import ThreadDist  # noqa: E402


def run_ranks(world, make_and_run):
    dist = ThreadDist(world)
    res, errs, times = [None] * world, [None] * world, [0.0] * world
    ctxs = [edsparser_amd.Context(0) for _ in range(world)]

    def work(rank):
        dist.local.rank = rank
        try:
            dist.barrier.wait()
            t0 = time.perf_counter()
            res[rank] = make_and_run(ctxs[rank], rank, world, dist)
            times[rank] = time.perf_counter() - t0
        except Exception as ex:  # noqa: BLE001
            errs[rank] = ex
            dist.barrier.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if any(errs):
        raise [e for e in errs if e is not None][0]
    return res, max(times)


def main():
    eds_mb = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    vcf_k = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    ctx = edsparser_amd.Context(0)
    ctx.leds_merge(b"{A}{C,G}{T}", None, 10)
    t0 = time.perf_counter()
    eds, seds = genrandomeds_shaped(eds_mb, 0.10, 42)
    print("EDS %.1f MB + sEDS %.1f MB generated in %.1f s" % (len(eds) / 1e6, len(seds) / 1e6, time.perf_counter() - t0), flush=True)
    t0 = time.perf_counter()
    whole = ctx.leds_merge(eds, seds, 32, True)
    t_whole = time.perf_counter() - t0
    for world in (2, 4, 8):
        res, t = run_ranks(world, lambda c, r, w, d: mg.gpu_merge_sharder(c, r, w, d).run(eds, seds, 32, True))
        same = (b"".join(x["leds"] for x in res), b"".join(x["seds"] for x in res)) == whole
        print("merge LINEAR l=32, %.1f MB in: single call %.3f s | %d ranges on one GPU %.3f s (partitioned=%s, ranges=%d) equal=%s"
              % ((len(eds) + len(seds)) / 1e6, t_whole, world, t, res[0]["partitioned"], res[0]["ranges"], same), flush=True)
    t0 = time.perf_counter()
    vcf, fasta = gen_vcf(vcf_k * 100_000, vcf_k * 1000, 8, 7)
    print("VCF %.1f MB (%d records, 8 samples) + FASTA %.1f MB generated in %.1f s" % (len(vcf) / 1e6, vcf_k * 1000, len(fasta) / 1e6, time.perf_counter() - t0), flush=True)
    ctx.vcf_transform(vcf[:4096], fasta, 0)
    t0 = time.perf_counter()
    wv = ctx.vcf_transform(vcf, fasta, 0)
    t_whole = time.perf_counter() - t0
    for world in (2, 4, 8):
        res, t = run_ranks(world, lambda c, r, w, d: mg.gpu_vcf_sharder(c, r, w, d).run(vcf, fasta))
        same = (b"".join(x["eds"] for x in res), b"".join(x["seds"] for x in res), res[0]["stats"]) == wv
        moved = sum(x["moved_lines_bytes"] for x in res)
        print("vcf2eds %.1f MB in: single call %.3f s | %d ranges on one GPU %.3f s (lines moved between ranks: %d B) equal=%s"
              % ((len(vcf) + len(fasta)) / 1e6, t_whole, world, t, moved, same), flush=True)


if __name__ == "__main__":
    main()
