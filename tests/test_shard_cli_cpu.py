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
        if tool == "msa2eds":
            shard.run_msa2eds(a, rank, world, dist, _oracle_msa_sharder(mg, rank, world, dist))
        elif tool == "vcf2eds":
            vs = mg.VcfSharder(rank, world, dist, o.vcf_index, o.vcf_sort_order, o.vcf_range)
            shard.run_vcf2eds(a, rank, world, dist, vs, ms)
        else:
            shard.run_eds2leds(a, rank, world, dist, ms)
    finally:
        dist.destroy_process_group()


def _oracle_msa_sharder(mg, rank, world, dist):
    """MsaSharder with the oracle in place of the C ABI: slab text from oracle.msa, edges from the slab's rows."""
    from test_multigpu_cpu import anchors_of, edges_of, rows_of

    def slab_fn(image, n_rows, ncols):
        rows = rows_of(image)
        e, s = o.msa(image, 0)
        return e, s, edges_of(rows, 0, ncols, e, s), lambda a, n: b"".join(r[a:a + n] for r in rows)

    def leds_fn(image, n_rows, ncols, l):
        rows = rows_of(image)
        e, s = o.msa(image, l)
        return e, s, anchors_of(rows, l, e, s), lambda a, n: b"".join(r[a:a + n] for r in rows)
    return mg.MsaSharder(rank, world, dist, slab_fn, lambda m: o.msa(m, 0), lambda m, l: o.msa(m, l), leds_fn=leds_fn,
                         mini_leds_fn=lambda m, l: o.msa(m, l))


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


@pytest.mark.parametrize("seed,lw,l", [(1, None, 0), (2, 7, 0), (3, 60, 0), (4, None, 0), (5, 13, 0), (6, None, 3), (7, 60, 5),
                                       (8, None, 2), (9, 7, 40)])
def test_msa2eds_two_ranks_cut_their_column_slabs(tmp_path, seed, lw, l):
    """File-based msa2eds over two ranks: every rank maps its column slab of every row (wrapped and one-line rows,
    with and without a trailing newline), stitches, and writes its piece; l > 0: the boundary is recomputed between the
    nearest standalone common runs of both slabs, or - no such runs, l = 40 - rank 0 transforms the file."""
    from msa_cases import random_msa
    rng = random.Random(900 + seed)
    S, L = rng.randint(2, 9), rng.randint(30, 400)
    msa = random_msa(rng, S=S, L=L, lw=lw or 10 ** 6, trailing_newline=seed % 2 == 1, p_var=rng.choice([0.05, 0.3, 0.7]))
    (tmp_path / "a.msa").write_bytes(msa)
    _run("msa2eds", ["msa2eds", "-i", str(tmp_path / "a.msa")] + (["-l", str(l)] if l else []))
    want = o.msa(msa, l)
    base = "a_l%d" % l if l else "a"
    assert (tmp_path / (base + (".leds" if l else ".eds"))).read_bytes() == want[0]
    assert (tmp_path / (base + ".seds")).read_bytes() == want[1]


def test_msa_layout_and_slab_images():
    """The slab cut addr(s, c) = start[s] + c + c / line_width (msa_transforms.cpp:268-269) against plain row slicing."""
    from edsparser_amd import multigpu as mg
    from msa_cases import random_msa
    from test_multigpu_cpu import rows_of
    rng = random.Random(77)
    for it in range(200):
        S, L = rng.randint(2, 6), rng.randint(2, 90)
        lw = rng.choice([10 ** 6, 1, 3, 7, 60])
        msa = random_msa(rng, S=S, L=L, lw=lw, trailing_newline=rng.random() < 0.5)
        lay = mg.msa_layout(msa)
        assert lay is not None and len(lay[0]) == S and lay[3] == L, (it, lw)
        rows = rows_of(msa)
        c0 = rng.randint(0, L - 1)
        c1 = rng.randint(c0 + 1, L)
        assert mg.msa_slab_image(msa, lay, c0, c1) == b"".join(b">r\n" + r[c0:c1] + b"\n" for r in rows), (it, lw, c0, c1)
    assert mg.msa_layout(b"") is None and mg.msa_layout(b"ACGT\n") is None and mg.msa_layout(b">a\nACGT\n") is None
    assert mg.msa_layout(b">a\nACGT\n>b\nAC\n") is None            # ragged rows: left to the unpartitioned transform


# ---- a failure on one rank reaches every rank with its own wording (no rank is left in a collective) ----------
def _failing_worker(rank, world, port, what, q):
    import torch.distributed as dist
    from edsparser_amd import multigpu as mg
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    text, kind = "", ""
    try:
        if what == "vcf":
            def range_fn(lines, fasta, cur0, nxt):
                if rank == 1:
                    raise ValueError("Invalid FASTA: rank one's wording")
                return o.vcf_range(lines, fasta, cur0, nxt)
            rng = random.Random(3)
            ref = "".join(rng.choice("ACGT") for _ in range(2000))
            vcf, fasta = _vcf(ref, _random_records(rng, ref, 200, 4), 4)
            mg.VcfSharder(rank, world, dist, o.vcf_index, o.vcf_sort_order, range_fn).run(vcf, fasta)
        else:
            from msa_cases import random_msa
            inner = _oracle_msa_sharder(mg, rank, world, dist)

            def slab_fn(image, n_rows, ncols):
                if rank == 1:
                    raise RuntimeError("Invalid MSA: slab one's wording")
                return inner.slab_fn(image, n_rows, ncols)
            msa = random_msa(random.Random(5), S=4, L=400)
            mg.MsaSharder(rank, world, dist, slab_fn, inner.mini_fn, inner.whole_fn).run(msa, 0)
    except Exception as ex:  # noqa: BLE001
        text, kind = str(ex), type(ex).__name__
    finally:
        dist.destroy_process_group()
    q.put((rank, kind, text))


@pytest.mark.parametrize("what", ["vcf", "msa"])
def test_failure_on_rank_one_is_worded_on_every_rank(what):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, what, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (k, t)) for r, k, t in (q.get(timeout=120) for _ in range(2)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = "Invalid FASTA: rank one's wording" if what == "vcf" else "Invalid MSA: slab one's wording"
    assert got[1][1] == want and got[1][0] in ("ValueError", "RuntimeError")          # the failing rank: its own exception
    assert got[0] == ("RemoteRankError", want)                                          # its peer: the same text, no hang
