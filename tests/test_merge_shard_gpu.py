"""GPU tests of the symbol-range partition of the merge through the C ABI (edsx_leds_merge_range): simulated ranks
are threads with one context each on the box's single GPU; one test runs two real ranks (processes) sharing it."""
import os
import random
import sys
import threading

import pytest

import oracle_lib as o
from test_merge_shard_cpu import shaped_eds
from test_vcf_shard_cpu import ThreadDist, _free_port

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_sharded_gpu(eds, seds, l, compact, world):
    import edsparser_amd
    from edsparser_amd import multigpu as mg
    dist = ThreadDist(world)
    results, errors = [None] * world, [None] * world

    def work(rank):
        dist.local.rank = rank
        try:
            ctx = edsparser_amd.Context(0)
            results[rank] = mg.gpu_merge_sharder(ctx, rank, world, dist).run(eds, seds, l, compact)
        except Exception as ex:  # noqa: BLE001
            errors[rank] = ex
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    if any(errors):
        real = [e for e in errors if e is not None]
        raise ([e for e in real if isinstance(e, edsparser_amd.EdsxError)] or real)[0]
    return b"".join(r["leds"] for r in results), b"".join(r["seds"] for r in results), results[0]


def test_range_call_matches_the_oracle_range():
    import edsparser_amd
    ctx = edsparser_amd.Context(0)
    rng = random.Random(21)
    for it in range(40):
        l = rng.choice([2, 5, 10])
        linear = rng.random() < 0.6
        body, sbody = shaped_eds(rng, rng.randint(3, 40), l, rng.random() < 0.5, linear, collapse_frac=rng.choice([0.0, 0.3]))
        # make the first and last symbols sentinels: long single strings next to degenerate symbols
        eds = b"{" + b"A" * (l + 3) + b"}{C,G}" + body.rstrip(b"\n") + b"{C,G}{" + b"T" * (l + 2) + b"}"
        seds = (b"{0}{1}{2}" + sbody.rstrip(b"\n") + b"{3}{4}{0}") if linear else None
        for compact in (True, False):
            for h, t in ((True, True), (True, False), (False, True)):
                try:
                    want = o.merge_range(eds, seds, l, compact, h, t)
                except o.OracleError as ex:
                    with pytest.raises(edsparser_amd.EdsxError) as ei:
                        ctx.leds_merge_range(eds, seds, l, compact, h, t)
                    assert ei.value.message == str(ex)
                    continue
                got = ctx.leds_merge_range(eds, seds, l, compact, h, t)
                assert got[2:] == want[2:], (it, h, t)
                if want[2] and want[3]:
                    assert got == want, (it, compact, h, t, eds, seds)


@pytest.mark.parametrize("seed", range(4))
def test_shaped_inputs_sharded_equal_whole_on_gpu(seed):
    import edsparser_amd
    rng = random.Random(700 + seed)
    partitioned = 0
    for it in range(10):
        l = rng.choice([2, 5, 10, 32])
        linear = rng.random() < 0.6
        eds, seds = shaped_eds(rng, rng.randint(5, 400), l, rng.random() < 0.5, linear,
                               short_frac=rng.choice([0.0, 0.1, 0.4]), adj_frac=rng.choice([0.0, 0.1, 0.3]) if linear else 0.05,
                               collapse_frac=rng.choice([0.0, 0.0, 0.2]))
        compact = rng.random() < 0.5
        try:
            want = o.merge(eds, seds, l, compact)
        except o.OracleError as ex:
            with pytest.raises(edsparser_amd.EdsxError) as ei:
                run_sharded_gpu(eds, seds, l, compact, 3)
            assert ei.value.message == str(ex)
            continue
        for world in (2, 4):
            leds, sout, info = run_sharded_gpu(eds, seds, l, compact, world)
            assert (leds, sout) == want, (seed, it, world, info["why"])
            partitioned += info["partitioned"]
    assert partitioned > 5


def test_collapsing_neighbour_falls_back_on_gpu():
    unit = "{A,C}GGGGGGGG"
    eds = (unit * 8 + "{C,G}" + "TTTTTTTT" + "{A,T}{G,C}{A,C}" + "GGGGGGGG{A,C}" * 5).encode()
    seds = ("{1}{2}{0}" * 8 + "{1}{2}" + "{0}" + "{1}{2}{1}{3}{1}{2}" + "{0}{1}{2}" * 5).encode()
    want = o.merge(eds, seds, 4, True)
    leds, sout, info = run_sharded_gpu(eds, seds, 4, True, 2)
    assert (leds, sout) == want and not info["partitioned"]


def test_bench_shape_sharded_equals_whole():
    """BASELINE configs[2] shape at 1/10 scale (10 Mb reference, 10 % sites, LINEAR, l = 32): four symbol ranges must
    concatenate to the single-call merge (itself checked against the oracle here)."""
    import edsparser_amd
    from merge_cases import genrandomeds_shaped
    eds, seds = genrandomeds_shaped(10, 0.10, 3)
    ctx = edsparser_amd.Context(0)
    whole = ctx.leds_merge(eds, seds, 32, True)
    assert whole == o.merge(eds, seds, 32, True)
    leds, sout, info = run_sharded_gpu(eds, seds, 32, True, 4)
    assert info["partitioned"] and info["ranges"] == 4
    assert (leds, sout) == whole


def _worker(rank, world, port, eds, seds, l, q):
    import torch.distributed as dist
    import edsparser_amd
    from edsparser_amd import multigpu as mg
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        ctx = edsparser_amd.Context(0)
        res = mg.gpu_merge_sharder(ctx, rank, world, dist).run(eds, seds, l, True)
        gathered = [None] * world
        dist.all_gather_object(gathered, (res["leds"], res["seds"], res["partitioned"]))
        if rank == 0:
            q.put((b"".join(g[0] for g in gathered), b"".join(g[1] for g in gathered), all(g[2] for g in gathered)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_the_gpu():
    import torch.multiprocessing as mp
    rng = random.Random(13)
    eds, seds = shaped_eds(rng, 3000, 16, True, True)
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, eds, seds, 16, q)) for r in range(2)]
    for p in procs:
        p.start()
    leds, sout, partitioned = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert (leds, sout) == o.merge(eds, seds, 16, True) and partitioned
