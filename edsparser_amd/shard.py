"""Multi-GPU front end of msa2eds / vcf2eds / eds2leds: one process per GPU under torch.distributed.run, inputs
memory-mapped, every rank writes its piece of the output files at its global offset (no gather through one rank).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m edsparser_amd.shard \\
        msa2eds -i x.msa [-o out.eds] [-s out.seds] [-l CONTEXT]
    python -m torch.distributed.run ... -m edsparser_amd.shard vcf2eds -i x.vcf -r ref.fa [-o out.eds] [-s out.seds] [-l CONTEXT]
    python -m torch.distributed.run ... -m edsparser_amd.shard eds2leds -i x.eds -l CONTEXT [-s x.seds] [-o out.leds] [--full]

Flags and default file names are the single-GPU CLIs' (reference msa2eds.cpp:35-41,139-160; vcf2eds.cpp:36-43,159-180;
eds2leds.cpp:38-46,161-162).  The partitions are multigpu.MsaSharder (column slabs of every row + the boundary stitch),
multigpu.VcfSharder (reference-position ranges) and multigpu.MergeSharder (symbol ranges).
"""
import argparse
import mmap
import os
import sys

from . import multigpu as mg


def map_file(path):
    """Read-only memory map (bytes-like: len, slicing, find/rfind); an empty file maps to b""."""
    with open(path, "rb") as f:
        if os.fstat(f.fileno()).st_size == 0:
            return b""
        return mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)


def write_piece(path, data, offset, total, rank, dist):
    """Rank 0 sizes the file, then every rank writes its piece at its offset."""
    if rank == 0:
        with open(path, "wb") as f:
            f.truncate(total)
    dist.barrier()
    if data:
        fd = os.open(path, os.O_WRONLY)
        try:
            os.pwrite(fd, data, offset)
        finally:
            os.close(fd)
    dist.barrier()


def default_vcf_outputs(a):
    """vcf2eds.cpp:159-180"""
    stem = os.path.splitext(os.path.basename(a.input))[0]
    d = os.path.dirname(a.input)
    if a.context_length > 0:
        suffix = "_l%d" % a.context_length
        eds = a.output or os.path.join(d, stem + suffix + ".leds")
        seds = a.sources or os.path.join(os.path.dirname(eds), stem + suffix + ".seds")
    else:
        eds = a.output or os.path.join(d, stem + ".eds")
        seds = a.sources or os.path.join(os.path.dirname(eds), os.path.splitext(os.path.basename(eds))[0] + ".seds")
    return eds, seds


def run_vcf2eds(a, rank, world, dist, vcf_sharder, merge_sharder):
    vcf, fasta = map_file(a.input), map_file(a.reference)
    eds_path, seds_path = default_vcf_outputs(a)
    res = vcf_sharder.run(vcf, fasta)
    eds, seds = res["eds"], res["seds"]
    e_off, e_tot, s_off, s_tot = res["eds_offset"], res["eds_total"], res["seds_offset"], res["seds_total"]
    if a.context_length > 0:
        # vcf_transforms.cpp:735-755: the EDS text goes through the LINEAR merge (compact).  The merge partition
        # scans byte ranges of the whole text, so the pieces are gathered first.
        comm = mg.TensorComm(dist, world)
        all_eds, all_seds = comm.payloads(eds), comm.payloads(seds)
        m = merge_sharder.run(b"".join(all_eds), b"".join(all_seds), a.context_length, True)
        eds, seds = m["leds"], m["seds"]
        e_off, e_tot, s_off, s_tot = m["leds_offset"], m["leds_total"], m["seds_offset"], m["seds_total"]
    write_piece(eds_path, eds, e_off, e_tot, rank, dist)
    write_piece(seds_path, seds, s_off, s_tot, rank, dist)
    return eds_path, seds_path, res["stats"]


def run_msa2eds(a, rank, world, dist, msa_sharder):
    if os.path.splitext(a.input)[1] != ".msa":
        raise SystemExit("Error: Input file must be an MSA file (.msa)")
    msa = map_file(a.input)
    stem, d = os.path.splitext(os.path.basename(a.input))[0], os.path.dirname(a.input)
    if a.context_length > 0:                                   # msa2eds.cpp:139-160
        suffix = "_l%d" % a.context_length
        eds_path = a.output or os.path.join(d, stem + suffix + ".leds")
        seds_path = a.sources or os.path.join(os.path.dirname(eds_path), stem + suffix + ".seds")
    else:
        eds_path = a.output or os.path.join(d, stem + ".eds")
        seds_path = a.sources or os.path.join(os.path.dirname(eds_path), os.path.splitext(os.path.basename(eds_path))[0] + ".seds")
    res = msa_sharder.run(msa, a.context_length)
    write_piece(eds_path, res["eds"], res["eds_offset"], res["eds_total"], rank, dist)
    write_piece(seds_path, res["seds"], res["seds_offset"], res["seds_total"], rank, dist)
    return eds_path, seds_path, res


def run_eds2leds(a, rank, world, dist, merge_sharder):
    if a.context_length <= 0:
        raise SystemExit("Error: context_length must be > 0 for l-EDS transformation")
    eds = map_file(a.input)
    seds = map_file(a.sources) if a.sources else None
    stem = os.path.splitext(os.path.basename(a.input))[0]
    out = a.output or os.path.join(os.path.dirname(a.input), "%s_l%d.leds" % (stem, a.context_length))   # eds2leds.cpp:161-162
    m = merge_sharder.run(eds, seds, a.context_length, not a.full)
    write_piece(out, m["leds"], m["leds_offset"], m["leds_total"], rank, dist)
    sout = None
    if seds is not None:
        sout = os.path.splitext(out)[0] + ".seds"
        write_piece(sout, m["seds"], m["seds_offset"], m["seds_total"], rank, dist)
    return out, sout, m


def parse(argv):
    ap = argparse.ArgumentParser(prog="edsparser_amd.shard")
    sub = ap.add_subparsers(dest="tool", required=True)
    m = sub.add_parser("msa2eds")
    m.add_argument("-i", "--input", required=True)
    m.add_argument("-o", "--output", default="")
    m.add_argument("-s", "--sources", default="")
    m.add_argument("-l", "--context-length", type=int, default=0)
    v = sub.add_parser("vcf2eds")
    v.add_argument("-i", "--input", required=True)
    v.add_argument("-r", "--reference", required=True)
    v.add_argument("-o", "--output", default="")
    v.add_argument("-s", "--sources", default="")
    v.add_argument("-l", "--context-length", type=int, default=0)
    e = sub.add_parser("eds2leds")
    e.add_argument("-i", "--input", required=True)
    e.add_argument("-l", "--context-length", type=int, required=True)
    e.add_argument("-s", "--sources", default="")
    e.add_argument("-o", "--output", default="")
    e.add_argument("--full", action="store_true")
    e.add_argument("-t", "--threads", type=int, default=1)      # accepted and ignored, like the single-GPU CLI
    return ap.parse_args(argv)


def main(argv=None):
    import torch
    import torch.distributed as dist
    from . import Context
    a = parse(sys.argv[1:] if argv is None else argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("EDSX_SHARE_GPU"):                       # rehearsal on a one-GPU box: every rank uses cuda:0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    backend = os.environ.get("EDSX_DIST_BACKEND", "nccl")      # nccl = RCCL
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    try:
        ctx = Context(local_rank)                              # fails loudly without a gfx950 device
        ms = mg.gpu_merge_sharder(ctx, rank, world, dist)
        if a.tool == "msa2eds":
            eds_path, seds_path, res = run_msa2eds(a, rank, world, dist, mg.gpu_msa_sharder(ctx, Context(local_rank), rank, world, dist))
            if rank == 0:
                print("Transformation complete!\n  Output: \"%s\"\n  Sources: \"%s\"\n  Column slabs: %d%s"
                      % (eds_path, seds_path, world if res["partitioned"] else 1, "" if res["partitioned"] else " (not partitioned)"))
        elif a.tool == "vcf2eds":
            eds_path, seds_path, stats = run_vcf2eds(a, rank, world, dist, mg.gpu_vcf_sharder(ctx, rank, world, dist), ms)
            if rank == 0:
                print("Transformation complete!\n  Output: \"%s\"\n  Sources: \"%s\"\n" % (eds_path, seds_path))
                print("Variant Processing Statistics:")
                print("  Total variants read:        %d" % stats["total_variants"])
                print("  Successfully processed:     %d" % stats["processed_variants"])
                print("  Skipped (malformed):        %d" % stats["skipped_malformed"])
                print("  Skipped (unsupported SV):   %d" % stats["skipped_unsupported_sv"])
                print("  Variant groups created:     %d" % stats["variant_groups"])
        else:
            out, sout, m = run_eds2leds(a, rank, world, dist, ms)
            if rank == 0:
                print("Transformation complete!\n  Output: \"%s\"%s\n  Symbol ranges: %d%s"
                      % (out, ("\n  Output sources: \"%s\"" % sout) if sout else "", m["ranges"],
                         "" if m["partitioned"] else " (not partitioned: %s)" % m["why"]))
    except Exception as ex:  # noqa: BLE001 — the CLIs print "Error: <what>" and exit 1
        if rank == 0:
            print("Error: %s" % (getattr(ex, "message", None) or ex), file=sys.stderr)
        dist.destroy_process_group()
        sys.exit(1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
