"""CPU tests of the multi-GPU column-slab stitch (edsparser_amd/multigpu.py).  The slab transforms
come from the oracle here; on the GPU box the same stitcher is wired to the C ABI.
  * plan logic on 2..6 simulated slabs, including slabs that are a single run (chains)
  * the real SlabStitcher over torch.distributed gloo, world_size 2
"""
import os
import random
import socket
import sys

import pytest

import oracle_lib as o
from msa_cases import random_msa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edsparser_amd import multigpu as mg  # noqa: E402


def rows_of(msa):
    rows, cur = [], None
    for line in msa.split(b"\n"):
        if line.startswith(b">"):
            cur = bytearray()
            rows.append(cur)
        elif cur is not None:
            cur += line
    return [bytes(r) for r in rows]


def slab_image(rows, c0, c1):
    return b"".join(b">r\n" + r[c0:c1] + b"\n" for r in rows)


def edges_of(rows, c0, c1, eds, seds):
    """SlabEdges from the slab's rows and its oracle output (what the C ABI reports on the GPU)."""
    L = c1 - c0
    var = [any(r[c] != rows[0][c] for r in rows) or rows[0][c:c + 1] == b"-" for c in range(c0, c1)]
    runs = []
    for c in range(L):
        if c == 0 or var[c] != var[c - 1]:
            runs.append([var[c], 0])
        runs[-1][1] += 1
    # text of the first / last symbol
    first_e = eds.index(b"}") + 1
    last_e = len(eds) - eds.rindex(b"{")
    k_first = eds[:first_e].count(b",") + 1
    k_last = eds[len(eds) - last_e:].count(b",") + 1

    def seds_prefix(k):
        pos = 0
        for _ in range(k):
            pos = seds.index(b"}", pos) + 1
        return pos

    def seds_suffix(k):
        pos = len(seds)
        for _ in range(k):
            pos = seds.rindex(b"{", 0, pos)
        return len(seds) - pos
    return mg.SlabEdges(n_segments=len(runs), cols=L, eds_bytes=len(eds), seds_bytes=len(seds),
                        first_is_variant=int(runs[0][0]), first_cols=runs[0][1], first_eds_bytes=first_e,
                        first_seds_bytes=seds_prefix(k_first), last_is_variant=int(runs[-1][0]), last_cols=runs[-1][1],
                        last_eds_bytes=last_e, last_seds_bytes=seds_suffix(k_last))


def anchors_of(rows, l, eds, seds):
    """edsx_msa_anchor_info on the CPU (what MsaSharder's l-EDS path needs from a slab): the first and the last common
    segment of at least l columns, from the slab's rows (segments by the l-EDS rule, msa_transforms.cpp:133-190) and
    the slab's oracle output (text offsets: segment i is the i-th brace group of the .eds and owns one source set per
    string)."""
    L = len(rows[0])
    common = [all(r[c] == rows[0][c] for r in rows) and rows[0][c:c + 1] != b"-" for c in range(L)]
    starts, i, prev_standalone = [], 0, False
    while i < L:
        e = i
        while e < L and common[e] == common[i]:
            e += 1
        if common[i]:
            if (e - i) >= l or i == 0 or e == L:
                starts.append(i)
                prev_standalone = True
            else:
                if prev_standalone:
                    starts.append(i)
                prev_standalone = False
        elif prev_standalone:
            starts.append(i)
            prev_standalone = False
        i = e
    if not starts or starts[0] != 0:
        starts = [0] + starts
    bounds = starts + [L]
    eoff, soff, epos, spos = [], [], 0, 0
    for _ in range(len(starts)):
        eoff.append(epos)
        soff.append(spos)
        close = eds.index(b"}", epos)
        for _k in range(eds[epos:close].count(b",") + 1):
            spos = seds.index(b"}", spos) + 1
        epos = close + 1
    eoff.append(len(eds))
    soff.append(len(seds))
    assert epos == len(eds) and spos == len(seds)
    anchors = [k for k in range(len(starts)) if common[starts[k]] and bounds[k + 1] - bounds[k] >= l]
    if not anchors:
        return {"n_segments": len(starts), "found": 0, "first_seg": 0, "last_seg": 0, "last_col": 0, "last_eds_bytes": 0,
                "last_seds_bytes": 0, "first_end": 0, "first_eds_end": 0, "first_seds_end": 0}
    f, t = anchors[0], anchors[-1]
    return {"n_segments": len(starts), "found": 1, "first_seg": f, "last_seg": t, "last_col": bounds[t], "last_eds_bytes": eoff[t],
            "last_seds_bytes": soff[t], "first_end": bounds[f + 1], "first_eds_end": eoff[f + 1], "first_seds_end": soff[f + 1]}


def stitch_serial(rows, cuts):
    """All ranks simulated in one process with the library's planning functions."""
    S = len(rows)
    bounds = [0] + list(cuts) + [len(rows[0])]
    slabs = []
    for i in range(len(bounds) - 1):
        e, s = o.msa(slab_image(rows, bounds[i], bounds[i + 1]), 0)
        slabs.append((bounds[i], bounds[i + 1], e, s))
    edges = [edges_of(rows, c0, c1, e, s) for c0, c1, e, s in slabs]
    plan = mg.plan_stitch(edges)
    eds_out, seds_out = b"", b""
    for r, (c0, c1, e, s) in enumerate(slabs):
        act = plan.actions[r]
        for ci in act.owns:
            ch = plan.chains[ci]
            blocks = []
            for q in range(ch.first, ch.last + 1):
                a, n = mg.chain_columns(ch, q, edges)
                q0 = slabs[q][0]
                blocks.append((b"".join(row[q0 + a:q0 + a + n] for row in rows), n))
            me, ms = o.msa(mg.mini_alignment(blocks, S), 0)
            act.extra_eds += me
            act.extra_seds += ms
        e_lo, e_hi, s_lo, s_hi = mg.piece_bounds(edges[r], act)
        eds_out += e[e_lo:e_hi] + act.extra_eds
        seds_out += s[s_lo:s_hi] + act.extra_seds
    return eds_out, seds_out


def test_plan_on_simulated_slabs():
    rng = random.Random(4242)
    checked = 0
    for it in range(300):
        msa = random_msa(rng, S=rng.randint(2, 6), L=rng.randint(6, 60), lw=10 ** 6, p_var=rng.choice([0.08, 0.3, 0.6]))
        rows = rows_of(msa)
        L = len(rows[0])
        nslab = rng.randint(2, min(6, L))
        cuts = sorted(rng.sample(range(1, L), nslab - 1))
        want = o.msa(msa, 0)
        assert stitch_serial(rows, cuts) == want, (it, cuts, msa)
        checked += 1
    assert checked == 300


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, msa, cut, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        rows = rows_of(msa)
        L = len(rows[0])
        bounds = [0, cut, L]
        c0, c1 = bounds[rank], bounds[rank + 1]
        e, s = o.msa(slab_image(rows, c0, c1), 0)
        st = mg.SlabStitcher(rank, world, len(rows), dist, lambda m: o.msa(m, 0), lambda: edges_of(rows, c0, c1, e, s),
                             lambda a, n: b"".join(r[c0 + a:c0 + a + n] for r in rows))
        res = st.stitch()
        piece = mg.stitched_piece(e, s, res)
        gathered = [None] * world
        dist.all_gather_object(gathered, (res["eds_offset"], res["seds_offset"], piece))
        if rank == 0:
            eds = b"".join(p[2][0] for p in gathered)
            seds = b"".join(p[2][1] for p in gathered)
            offs_ok = all(gathered[i][0] == sum(len(gathered[j][2][0]) for j in range(i)) for i in range(world))
            q.put((eds, seds, offs_ok, res["eds_total"], res["seds_total"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_stitcher_over_gloo_world2(seed):
    import torch.multiprocessing as mp
    rng = random.Random(seed)
    ctx = mp.get_context("spawn")
    for _ in range(3):
        msa = random_msa(rng, S=rng.randint(2, 8), L=rng.randint(20, 120), lw=10 ** 6, p_var=rng.choice([0.1, 0.4]))
        L = len(rows_of(msa)[0])
        cut = rng.randint(1, L - 1)
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, msa, cut, q)) for r in range(2)]
        for p in procs:
            p.start()
        eds, seds, offs_ok, etot, stot = q.get(timeout=120)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        want = o.msa(msa, 0)
        assert (eds, seds) == want, (cut, msa)
        assert offs_ok and etot == len(eds) and stot == len(seds)
