"""GPU test of the slab stitch wired to the C ABI: two ranks share the one GPU of the test box
(gloo carries the small object collectives), each transforms its own synthetic column slab, the
stitched concatenation must equal the single-GPU transform of the whole alignment."""
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, S, Lr, seed, q, vfrac=0.3):
    import torch
    import torch.distributed as dist
    import edsparser_amd
    from edsparser_amd import multigpu as mg
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        ctx = edsparser_amd.Context(0)
        mini = edsparser_amd.Context(0)
        n = edsparser_amd.synth_size(S, Lr)
        buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
        ctx.msa_synth_device(buf.data_ptr(), n, S, Lr, col0=rank * Lr, variant_fraction=vfrac, seed=seed)
        E, Q = ctx.msa_plan_device(buf.data_ptr(), n, 0)
        d_e = torch.empty(E + 16, dtype=torch.uint8, device="cuda:0")
        d_s = torch.empty(Q + 16, dtype=torch.uint8, device="cuda:0")
        ctx.msa_emit_device(d_e.data_ptr(), d_s.data_ptr())
        torch.cuda.synchronize()
        st = mg.gpu_stitcher(ctx, mini, rank, world, S, Lr, dist)
        st.set_sizes(E, Q)
        res = st.stitch()
        piece = mg.stitched_piece(bytes(d_e[:E].cpu().numpy()), bytes(d_s[:Q].cpu().numpy()), res)
        gathered = [None] * world
        dist.all_gather_object(gathered, piece)
        if rank == 0:
            q.put((b"".join(p[0] for p in gathered), b"".join(p[1] for p in gathered), res["chains"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("S,Lr,seed", [(12, 3000, 1), (50, 20000, 2), (1000, 700, 3)])
def test_two_slabs_on_one_gpu(S, Lr, seed):
    import torch
    import torch.multiprocessing as mp
    import edsparser_amd
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, S, Lr, seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    eds, seds, chains = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ctx = edsparser_amd.Context(0)
    n = edsparser_amd.synth_size(S, 2 * Lr)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    ctx.msa_synth_device(buf.data_ptr(), n, S, 2 * Lr, col0=0, variant_fraction=0.3, seed=seed)
    torch.cuda.synchronize()
    want = ctx.msa_transform(bytes(buf.cpu().numpy()), 0)
    assert (eds, seds) == want


def test_bench_shape_slabs_vs_whole():
    """The bench workload's shape at 1/10 of its width (1000 rows x 10 M columns, 5 % sites, 10 GB): two
    column slabs transformed by two ranks and stitched must give, byte for byte, the single-GPU
    transform of the whole alignment (1.7 GB of .seds).  A size-independent property: it exercises
    every kernel of the fast path at the bench's row count and the boundary stitch, without the oracle."""
    import hashlib
    import torch
    import torch.multiprocessing as mp
    import edsparser_amd
    S, Lr, seed = 1000, 5_000_000, 42
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, S, Lr, seed, q, 0.05)) for r in range(2)]
    for p in procs:
        p.start()
    eds, seds, chains = q.get(timeout=600)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    ctx = edsparser_amd.Context(0)
    n = edsparser_amd.synth_size(S, 2 * Lr)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    ctx.msa_synth_device(buf.data_ptr(), n, S, 2 * Lr, col0=0, variant_fraction=0.05, seed=seed)
    E, Q = ctx.msa_plan_device(buf.data_ptr(), n, 0)
    d_e = torch.empty(E + 16, dtype=torch.uint8, device="cuda:0")
    d_s = torch.empty(Q + 16, dtype=torch.uint8, device="cuda:0")
    ctx.msa_emit_device(d_e.data_ptr(), d_s.data_ptr())
    torch.cuda.synchronize()
    assert (len(eds), len(seds)) == (E, Q)
    assert hashlib.sha256(eds).digest() == hashlib.sha256(d_e[:E].cpu().numpy().tobytes()).digest()
    assert hashlib.sha256(seds).digest() == hashlib.sha256(d_s[:Q].cpu().numpy().tobytes()).digest()
