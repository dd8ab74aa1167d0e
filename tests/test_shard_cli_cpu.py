"""CPU test of the multi-GPU front end (edsparser_amd/shard.py) over gloo, world_size 2: memory-mapped inputs,
pieces written at their offsets; the sharders are wired to the oracle here (the C ABI needs a GPU)."""
import os
import random
import sys

import pytest

import oracle_lib as o
from test_merge_shard_cpu import shaped_eds
from test_vcf_shard_cpu import _free_port, _random_records, _vcf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, tool, argv):
    import torch.distributed as dist
    from edsparser_amd import multigpu as mg
    from edsparser_amd import shard
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        a = shard.parse(argv)
        ms = mg.MergeSharder(rank, world, dist, lambda e, s, l, c, h, t: o.merge_range(e, s, l, c, h, t),
                             lambda e, s, l, c: o.merge(e, s, l, c))
        if tool == "vcf2eds":
            vs = mg.VcfSharder(rank, world, dist, o.vcf_index, o.vcf_sort_order, o.vcf_range)
            shard.run_vcf2eds(a, rank, world, dist, vs, ms)
        else:
            shard.run_eds2leds(a, rank, world, dist, ms)
    finally:
        dist.destroy_process_group()


def _run(tool, argv):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tool, argv)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0


@pytest.mark.parametrize("l", [0, 6])
def test_vcf2eds_two_ranks_write_their_pieces(tmp_path, l):
    rng = random.Random(31 + l)
    ref = "".join(rng.choice("ACGT") for _ in range(4000))
    recs = _random_records(rng, ref, 300, 3)
    vcf, fasta = _vcf(ref, recs, 3)
    (tmp_path / "x.vcf").write_bytes(vcf)
    (tmp_path / "ref.fa").write_bytes(fasta)
    argv = ["vcf2eds", "-i", str(tmp_path / "x.vcf"), "-r", str(tmp_path / "ref.fa")] + (["-l", str(l)] if l else [])
    _run("vcf2eds", argv)
    want = o.vcf(vcf, fasta, l)
    base = "x_l%d" % l if l else "x"
    assert (tmp_path / (base + (".leds" if l else ".eds"))).read_bytes() == want[0]
    assert (tmp_path / (base + ".seds")).read_bytes() == want[1]


@pytest.mark.parametrize("linear", [False, True])
def test_eds2leds_two_ranks_write_their_pieces(tmp_path, linear):
    rng = random.Random(41)
    eds, seds = shaped_eds(rng, 150, 8, False, linear)
    (tmp_path / "g.eds").write_bytes(eds)
    argv = ["eds2leds", "-i", str(tmp_path / "g.eds"), "-l", "8"]
    if linear:
        (tmp_path / "g.seds").write_bytes(seds)
        argv += ["-s", str(tmp_path / "g.seds")]
    _run("eds2leds", argv)
    want = o.merge(eds, seds, 8, True)
    assert (tmp_path / "g_l8.leds").read_bytes() == want[0]
    if linear:
        assert (tmp_path / "g_l8.seds").read_bytes() == want[1]
