"""CPU tests of the multi-GPU symbol-range partition of the merge (edsparser_amd/multigpu.py, MergeSharder).
range_fn / whole_fn come from the oracle here; on the GPU box the same sharder is wired to the C ABI
(tests/test_merge_shard_gpu.py).  Expected output is always the unpartitioned merge of the whole text
(oracle, itself pinned on the reference's fixtures), plus the reference-generated fixtures directly."""
import json
import os
import random
import sys
import threading

import pytest

import oracle_lib as o
from conftest import GOLDEN
from test_vcf_shard_cpu import ThreadDist, _free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edsparser_amd import multigpu as mg  # noqa: E402


def _range_fn(e, s, l, c, h, t):
    return o.merge_range(e, s, l, c, h, t)


def _whole_fn(e, s, l, c):
    return o.merge(e, s, l, c)


def run_sharded(eds, seds, l, compact, world, range_fn=_range_fn, whole_fn=_whole_fn):
    dist = ThreadDist(world)
    results, errors = [None] * world, [None] * world

    def work(rank):
        dist.local.rank = rank
        try:
            results[rank] = mg.MergeSharder(rank, world, dist, range_fn, whole_fn).run(eds, seds, l, compact)
        except Exception as ex:  # noqa: BLE001
            errors[rank] = ex
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    if any(errors):
        assert all(e is not None for e in errors), errors
        raise errors[0]
    leds = b"".join(r["leds"] for r in results)
    sout = b"".join(r["seds"] for r in results)
    for r in range(world):
        assert results[r]["leds_offset"] == sum(len(results[q]["leds"]) for q in range(r))
        assert results[r]["seds_offset"] == sum(len(results[q]["seds"]) for q in range(r))
        assert results[r]["leds_total"] == len(leds) and results[r]["seds_total"] == len(sout)
    return leds, sout, results[0]


def shaped_eds(rng, nsites, l, compact_in, linear, short_frac=0.1, adj_frac=0.1, collapse_frac=0.0, paths=4):
    """genrandomeds-shaped text: common blocks between degenerate sites; some blocks shorter than l, some sites
    adjacent, optionally source sets that make products collapse to one string."""
    eds, seds = [], []

    def common(n):
        t = "".join(rng.choice("ACGT") for _ in range(n))
        eds.append(t if (compact_in and t) else "{" + t + "}")
        seds.append("{0}")

    def site():
        k = rng.randint(2, 4)
        alts = ["".join(rng.choice("ACGT") for _ in range(rng.choice([0, 1, 1, 1, 3]))) for _ in range(k)]
        eds.append("{" + ",".join(alts) + "}")
        if rng.random() < collapse_frac:
            for a in range(k):
                seds.append("{%d}" % rng.randint(1, paths))
        else:
            choice = [rng.randrange(k) for _ in range(paths)]
            for a in range(k):
                ids = [str(p + 1) for p in range(paths) if choice[p] == a]
                seds.append("{" + ",".join(ids or [str(rng.randint(1, paths))]) + "}")
    common(rng.randint(1, 3 * l))
    for _ in range(nsites):
        site()
        while rng.random() < adj_frac:
            site()
        common(rng.randint(0, l - 1) if rng.random() < short_frac else rng.randint(l, 4 * l))
    return "".join(eds).encode() + b"\n", ("".join(seds).encode() + b"\n") if linear else None


@pytest.mark.parametrize("seed", range(8))
def test_shaped_inputs_sharded_equal_whole(seed):
    rng = random.Random(500 + seed)
    partitioned = 0
    for it in range(12):
        l = rng.choice([2, 5, 10, 32])
        linear = rng.random() < 0.6
        eds, seds = shaped_eds(rng, rng.randint(5, 120), l, rng.random() < 0.5, linear,
                               short_frac=rng.choice([0.0, 0.1, 0.4]), adj_frac=rng.choice([0.0, 0.1, 0.3]) if linear else 0.05,
                               collapse_frac=rng.choice([0.0, 0.0, 0.2]))
        compact = rng.random() < 0.5
        try:
            want = o.merge(eds, seds, l, compact)
        except o.OracleError as ex:
            for world in (2, 3):
                with pytest.raises(o.OracleError) as ei:
                    run_sharded(eds, seds, l, compact, world)
                assert str(ei.value) == str(ex)
            continue
        for world in (2, 3, 5, 8):
            leds, sout, info = run_sharded(eds, seds, l, compact, world)
            assert (leds, sout) == want, (seed, it, world, info["why"], eds, seds, l, compact)
            partitioned += info["partitioned"]
    assert partitioned > 10


def test_collapsing_neighbour_pulls_the_sentinel_in():
    unit = "{A,C}GGGGGGGG"
    eds = (unit * 8 + "{C,G}" + "TTTTTTTT" + "{A,T}{G,C}{A,C}" + "GGGGGGGG{A,C}" * 5).encode()
    su = "{1}{2}{0}"
    seds = (su * 8 + "{1}{2}" + "{0}" + "{1}{2}{1}{3}{1}{2}" + "{0}{1}{2}" * 5).encode()
    want = o.merge(eds, seds, 4, True)
    assert b"TTTTTTTTAG" in want[0]
    leds, sout, info = run_sharded(eds, seds, 4, True, 2)
    assert (leds, sout) == want
    assert not info["partitioned"] and info["why"].startswith("a sentinel was merged")
    # the same text without sources (CARTESIAN products never collapse) is partitioned
    leds, sout, info = run_sharded(eds, None, 4, True, 2)
    assert (leds, sout) == o.merge(eds, None, 4, True) and info["partitioned"]


def test_reference_generated_fixtures_sharded():
    with open(os.path.join(GOLDEN, "gen_merge.json")) as f:
        cases = json.load(f)["cases"]
    for i, c in enumerate(cases):
        eds = c["eds"].encode()
        seds = c["seds"].encode() if c.get("seds") is not None else None
        world = 2 + i % 3
        if "error" in c["expect"]:
            with pytest.raises(o.OracleError) as ei:
                run_sharded(eds, seds, c["l"], c["compact"], world)
            assert str(ei.value) == c["expect"]["error"], (i, c)
            continue
        leds, sout, _ = run_sharded(eds, seds, c["l"], c["compact"], world)
        assert leds.decode() == c["expect"]["out"] and sout.decode() == c["expect"]["seds_out"], (i, c)


def test_unpartitionable_text_goes_whole():
    for eds, seds in [(b"{A,C} GGGG {T,G}\nAAAA{C,T}", None), (b"", None), (b"ACGT", None), (b"{A,C}{G,T}", b"{1}{2}{1}{2}"),
                      (b"{A,C}GGGG{T,G}AAAA{C,T}GGGG{A,C}", b"{1}{2}{0}{1}{2}{0}{1}{2}{0}")]:
        try:
            want = o.merge(eds, seds, 3, True)
        except o.OracleError as ex:
            with pytest.raises(o.OracleError) as ei:
                run_sharded(eds, seds, 3, True, 3)
            assert str(ei.value) == str(ex)
            continue
        assert run_sharded(eds, seds, 3, True, 3)[:2] == want


def _worker(rank, world, port, eds, seds, l, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        res = mg.MergeSharder(rank, world, dist, _range_fn, _whole_fn).run(eds, seds, l, True)
        gathered = [None] * world
        dist.all_gather_object(gathered, (res["leds"], res["seds"], res["partitioned"]))
        if rank == 0:
            q.put((b"".join(g[0] for g in gathered), b"".join(g[1] for g in gathered), all(g[2] for g in gathered)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("linear", [False, True])
def test_sharder_over_gloo_world2(linear):
    import torch.multiprocessing as mp
    rng = random.Random(9)
    eds, seds = shaped_eds(rng, 200, 8, True, linear)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, eds, seds, 8, q)) for r in range(2)]
    for p in procs:
        p.start()
    leds, sout, partitioned = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert (leds, sout) == o.merge(eds, seds, 8, True)
    assert partitioned
