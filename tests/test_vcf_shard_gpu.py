"""GPU tests of the position-range partition of the VCF path, through the C ABI (edsx_vcf_index,
edsx_vcf_sort_order, edsx_vcf_transform_range): simulated ranks are threads with one context each on the
box's single GPU; one test runs two real ranks (processes) that share the GPU.  Expected outputs: the
reference-generated fixtures, the oracle, and the unpartitioned edsx_vcf_transform at a larger size."""
import json
import os
import random
import sys
import threading

import pytest

import oracle_lib as o
from conftest import GOLDEN
from test_vcf_shard_cpu import ThreadDist, _free_port, _random_records, _vcf

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_sharded_gpu(vcf, fasta, world):
    import edsparser_amd
    from edsparser_amd import multigpu as mg
    dist = ThreadDist(world)
    results, errors = [None] * world, [None] * world

    def work(rank):
        dist.local.rank = rank
        try:
            ctx = edsparser_amd.Context(0)
            results[rank] = mg.gpu_vcf_sharder(ctx, rank, world, dist).run(vcf, fasta)
        except Exception as ex:  # noqa: BLE001
            errors[rank] = ex
            dist.barrier.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    if any(errors):
        real = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)]
        raise ([e for e in real if isinstance(e, edsparser_amd.EdsxError)] or real)[0]
    return (b"".join(r["eds"] for r in results), b"".join(r["seds"] for r in results), results[0]["stats"], results)


def test_index_and_sort_order_match_the_oracle():
    import edsparser_amd
    ctx = edsparser_amd.Context(0)
    rng = random.Random(3)
    ref = "".join(rng.choice("ACGT") for _ in range(3000))
    recs = _random_records(rng, ref, 400, 3, dup_frac=0.4)
    vcf, _ = _vcf(ref, recs, 3, shuffle=rng)
    vcf += b"badline\nc\t7\t.\tG\t<INV>\t.\t.\t.\tGT\t0|1\t0|0\t0|0\n"
    got, want = ctx.vcf_index(vcf), o.vcf_index(vcf)
    for a, b in zip(got[:4], want[:4]):
        assert a.tolist() == b.tolist()
    assert {k: v for k, v in got[4].items() if k != "variant_groups"} == {k: v for k, v in want[4].items() if k != "variant_groups"}
    assert ctx.vcf_sort_order(got[0]).tolist() == o.vcf_sort_order(want[0]).tolist()
    for off, ln in zip(got[2].tolist(), got[3].tolist()):
        assert vcf[off:off + ln].startswith(b"chr1\t")


def test_reference_generated_fixtures_sharded_on_gpu():
    import edsparser_amd
    with open(os.path.join(GOLDEN, "gen_vcf.json")) as f:
        cases = [c for c in json.load(f)["cases"] if c["l"] == 0]
    checked = 0
    for i, c in enumerate(cases):
        world = 2 + i % 3
        vcf, fasta = c["vcf"].encode(), c["fasta"].encode()
        if "error" in c["expect"]:
            with pytest.raises(edsparser_amd.EdsxError) as ei:
                run_sharded_gpu(vcf, fasta, world)
            assert ei.value.message == c["expect"]["error"]
            continue
        eds, seds, stats, _ = run_sharded_gpu(vcf, fasta, world)
        assert (eds.decode(), seds.decode()) == (c["expect"]["eds"], c["expect"]["seds"]), (i, world, c)
        assert stats == c["expect"]["stats"], (i, world)
        checked += 1
    assert checked > 150


@pytest.mark.parametrize("seed", range(3))
def test_random_sorted_shuffled_duplicates_vs_oracle(seed):
    rng = random.Random(2000 + seed)
    ref = "".join(rng.choice("ACGT") for _ in range(rng.randint(2000, 20000)))
    recs = _random_records(rng, ref, rng.randint(200, 1500), rng.choice([2, 8, 70]), dup_frac=[0.0, 0.3, 0.0][seed])
    for shuffle in (None, rng):
        vcf, fasta = _vcf(ref, recs, len(recs[0][3]), lw=rng.choice([60, 7]), shuffle=shuffle)
        want = o.vcf(vcf, fasta, 0)
        for world in (2, 4):
            assert run_sharded_gpu(vcf, fasta, world)[:3] == want, (seed, world)


def test_larger_sharded_equals_unpartitioned():
    """BASELINE configs[3] shape at 1/100 scale (10 Mb reference, 10^5 records, 8 diploid samples): four position
    ranges must concatenate to the single-call transform, which the oracle checks once."""
    import edsparser_amd
    rng = random.Random(11)
    L, n = 10_000_000, 100_000
    ref = "".join(rng.choices("ACGT", k=L))
    recs = _random_records(rng, ref, n, 8)
    vcf, fasta = _vcf(ref, recs, 8)
    ctx = edsparser_amd.Context(0)
    whole = ctx.vcf_transform(vcf, fasta, 0)
    assert whole == o.vcf(vcf, fasta, 0)
    eds, seds, stats, res = run_sharded_gpu(vcf, fasta, 4)
    assert (eds, seds, stats) == whole
    assert min(r["records"] for r in res) > n // 5                   # balanced ranges
    assert sum(r["moved_lines_bytes"] for r in res) < len(vcf) // 100   # sorted file: only slivers move


def _worker(rank, world, port, vcf, fasta, q):
    import torch.distributed as dist
    import edsparser_amd
    from edsparser_amd import multigpu as mg
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        ctx = edsparser_amd.Context(0)
        res = mg.gpu_vcf_sharder(ctx, rank, world, dist).run(vcf, fasta)
        gathered = [None] * world
        dist.all_gather_object(gathered, (res["eds"], res["seds"]))
        if rank == 0:
            q.put((b"".join(g[0] for g in gathered), b"".join(g[1] for g in gathered), res["stats"]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_the_gpu():
    import torch.multiprocessing as mp
    rng = random.Random(5)
    ref = "".join(rng.choice("ACGT") for _ in range(50000))
    recs = _random_records(rng, ref, 4000, 8, dup_frac=0.05)
    vcf, fasta = _vcf(ref, recs, 8)
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, vcf, fasta, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == o.vcf(vcf, fasta, 0)
