#!/usr/bin/env python3
"""bench.py — MSA -> EDS input MB/s on MI355X (BASELINE.json metric), with the HBM roofline of the
dominant kernel and a CPU baseline timed in the same run.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path (edsx_msa_plan_device + edsx_msa_emit_device: row index,
column scan + variant-column extraction, segment table, per-segment grouping, scans, .eds/.seds
text) over one synthetic alignment that is already resident in HBM.  Workload at N=1: BASELINE
configs[4], 1000 sequences x 100 Mb (genrandomeds-shaped, 5 % variant sites, one line per row).
With N ranks (--scaling strong, the default: BASELINE configs[4] is ONE alignment over 1/2/4/8 GPUs) the same
1000 x 100 Mb alignment is range-partitioned into N column slabs of 100/N Mb; every rank transforms its own slab
and the boundary segments are stitched over RCCL inside the step (fixed-size tensor collectives).
--scaling weak keeps 100 Mb columns per rank (an N x 100 Mb alignment).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md "HBM3E peak BW")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["c5", "c2"], default="c5",
                    help="c5: BASELINE configs[4], 1000 sequences x 100 Mb (the headline, default); c2: BASELINE configs[1], "
                         "64 sequences x 10 Mb (a parity-test shape; its line is kept on record under profiles/)")
    ap.add_argument("--rows", type=int, default=1000)
    ap.add_argument("--cols", type=int, default=100_000_000,
                    help="alignment columns (strong scaling: of the whole alignment; weak: per GPU)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: one alignment of --cols columns split into N column slabs; weak: --cols columns per rank")
    ap.add_argument("--context-len", type=int, default=0)
    ap.add_argument("--variant-fraction", type=float, default=0.05)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cpu-baseline-mb", type=float, default=3000.0,
                    help="size of the CPU-baseline sample in MB (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a one-GPU box: every rank uses cuda:0")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: take the multi-rank code path (process group, stitch, reductions) with WORLD_SIZE=1 too")
    ap.add_argument("--verify", type=int, default=1,
                    help="N=1, context length 0: after the timed region compare sampled column windows of the outputs "
                         "(spread over the whole width, incl. offsets beyond 4 GiB and the last columns) with the CPU "
                         "oracle (tests/fullsize_verify.py); 0 = skip")
    ap.add_argument("--row-align", type=int, default=0,
                    help="lay the synthetic rows out with blank-padded headers so that every row starts on a multiple of "
                         "this many bytes (0: the plain FASTA image - the headline workload)")
    ap.add_argument("--aligned-probe", type=int, default=128,
                    help="N=1: after the timed run repeat it on the same alignment laid out with rows on multiples of this many "
                         "bytes (what an upload that places the rows for the scan produces) and report it beside the headline "
                         "under \"aligned_rows\"; 0 = skip")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc run")
    a = ap.parse_args()
    if a.workload == "c2":
        a.rows, a.cols = 64, 10_000_000
        a.cpu_baseline_mb = min(a.cpu_baseline_mb, 640.0)
    return a


def cpu_baseline(ctx, torch, rows, sample_mb, vfrac, seed, l):
    """The oracle ("port" of the reference's 3-pass algorithm) on one host core, bounded sample of
    the same workload shape (same rows, fewer columns)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import edsparser_amd
    cols = max(1000, int(sample_mb * 1e6 / (rows + 1)))
    n = edsparser_amd.synth_size(rows, cols)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda")
    ctx.msa_synth_device(buf.data_ptr(), n, rows, cols, variant_fraction=vfrac, seed=seed)
    torch.cuda.synchronize()
    host = bytes(buf.cpu().numpy())
    del buf
    t0 = time.perf_counter()
    e, s = oracle_lib.msa(host, l)
    dt = time.perf_counter() - t0
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(n / dt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port", "cpu_model": model,
            "flavour": "oracle/msa_oracle.cpp on an in-memory image of the file (the reference's three passes without its "
                       "seek-per-sequence file reads: the faster of the two flavours BASELINE.md names)",
            "host_cores": os.cpu_count(),
            "sample": "%d rows x %d columns (%.0f MB in, %.1f s), same generator and site fraction"
                      % (rows, cols, n / 1e6, dt)}


def main():
    a = parse_args()
    import torch
    import torch.distributed as dist
    import edsparser_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run" % (a.gpus, world),
                  file=sys.stderr)
        sys.exit(2)
    if a.share_gpu:
        local_rank = 0
    distributed = world > 1 or a.force_dist
    torch.cuda.set_device(local_rank)
    if distributed:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend)

    ctx = edsparser_amd.Context(local_rank)
    S, l = a.rows, a.context_len
    if a.scaling == "strong":                                  # this rank's column slab of the one alignment
        col0, col1 = a.cols * rank // world, a.cols * (rank + 1) // world
    else:
        col0, col1 = a.cols * rank, a.cols * (rank + 1)
    L = col1 - col0
    n = edsparser_amd.synth_size(S, L, a.row_align)
    msa = torch.empty(n, dtype=torch.uint8, device="cuda")
    ctx.msa_synth_device(msa.data_ptr(), n, S, L, col0=col0, variant_fraction=a.variant_fraction,
                         seed=a.seed, row_align=a.row_align)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream

    stitcher = None
    if distributed:
        # boundary-segment stitch: KB-sized all_gather_object exchanges over RCCL (nccl backend)
        from edsparser_amd.multigpu import gpu_stitcher
        stitcher = gpu_stitcher(ctx, edsparser_amd.Context(local_rank), rank, world, S, L, dist)

    out = {"eds": None, "seds": None, "E": 0, "Q": 0}

    def step():
        E, Q = ctx.msa_plan_device(msa.data_ptr(), n, l, stream)
        if out["eds"] is None or out["eds"].numel() < E + 16:
            out["eds"] = torch.empty(E + 16, dtype=torch.uint8, device="cuda")
        if out["seds"] is None or out["seds"].numel() < Q + 16:
            out["seds"] = torch.empty(Q + 16, dtype=torch.uint8, device="cuda")
        ctx.msa_emit_device(out["eds"].data_ptr(), out["seds"].data_ptr(), stream)
        out["E"], out["Q"] = E, Q
        if stitcher is not None:
            stitcher.set_sizes(E, Q)
            out["stitch"] = stitcher.stitch()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if not os.environ.get("EDSX_BENCH_NOTIMING"):
        ctx.set_timing(True)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    timing = ctx.get_timing()
    ctx.set_timing(False)
    n_total = float(n)
    if distributed:
        dev = "cuda" if a.backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        t = torch.tensor([float(n), float(out["E"]), float(out["Q"])], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)              # input bytes of all slabs (and the slabs' own text sizes)
        n_total = float(t[0].item())

    info = ctx.msa_info()
    if rank == 0:
        total_in = n_total * a.steps
        value = total_in / dt / 1e6
        dom = max(timing, key=lambda t: t[1]) if timing else None
        roof = None
        per_kernel = {}
        for name, tot, cnt in timing:
            per_kernel[name] = round(tot / max(cnt, 1), 4)
        if dom:
            name, tot, cnt = dom
            avg_ms = tot / cnt
            if name == "k_scan_extract":
                alg = float(S) * float(L)     # 1 B read per alignment cell
                what = "S*L alignment cells read once"
            elif name in ("k_emit_variant", "k_seg_count"):
                alg = float(info["n_variant_cols"]) * S + (out["Q"] if name == "k_emit_variant" else 0)
                what = "variant-column bytes read" + (" + .seds bytes written" if name == "k_emit_variant" else "")
            else:
                alg = float(n)
                what = "input bytes"
            traffic = a.traffic_bytes
            if traffic is None:      # HBM bytes per launch from the committed rocprofv3 --pmc passes of this workload
                try:
                    tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
                    traffic = tj.get("%dx%d" % (S, L), {}).get(name)
                except OSError:
                    traffic = None
            ach = alg / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic,
                    "traffic_source": ("--traffic-bytes" if a.traffic_bytes is not None else
                                       "profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this build "
                                       "(profiles/collect.sh), not counted in this run") if traffic is not None else None,
                    "avg_kernel_ms": round(avg_ms, 4),
                    "algorithmic_bytes_per_launch": alg, "algorithmic_bytes": what}
        verify = None
        if world == 1 and a.verify and l == 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from fullsize_verify import verify_windows
            verify = verify_windows(ctx, torch, S, L, a.seed, a.variant_fraction, out["eds"], out["seds"],
                                    out["E"], out["Q"])
        aligned = None
        if world == 1 and a.aligned_probe > 1 and a.row_align != a.aligned_probe:
            # The same cells with every row on a multiple of --aligned-probe bytes (blank-padded headers).  The scan's 16-byte
            # loads are then aligned; a FASTA image as it comes (the headline above) has every row at its own odd offset.
            first = (out["eds"][:out["E"]].clone(), out["seds"][:out["Q"]].clone(), out["E"], out["Q"])
            del msa
            torch.cuda.empty_cache()
            n2 = edsparser_amd.synth_size(S, L, a.aligned_probe)
            msa2 = torch.empty(n2, dtype=torch.uint8, device="cuda")
            ctx.msa_synth_device(msa2.data_ptr(), n2, S, L, col0=col0, variant_fraction=a.variant_fraction, seed=a.seed,
                                 row_align=a.aligned_probe)
            torch.cuda.synchronize()

            def step2():
                E, Q = ctx.msa_plan_device(msa2.data_ptr(), n2, l, stream)
                ctx.msa_emit_device(out["eds"].data_ptr(), out["seds"].data_ptr(), stream)
                return E, Q
            step2()
            torch.cuda.synchronize()
            ctx.set_timing(True)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                E2, Q2 = step2()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            tm2 = {nm: tot / max(cnt, 1) for nm, tot, cnt in ctx.get_timing()}
            ctx.set_timing(False)
            same = (E2, Q2) == (first[2], first[3]) and bool(torch.equal(out["eds"][:E2], first[0])) and \
                bool(torch.equal(out["seds"][:Q2], first[1]))
            v2 = n2 * a.steps / dt2 / 1e6
            aligned = {"row_align": a.aligned_probe, "input_bytes": n2, "ms_per_step": round(dt2 / a.steps * 1e3, 3),
                       "value": round(v2, 1), "unit": "MB/s",
                       "frac_of_hbm_read_roofline": round(v2 * 1e6 / (HBM_PEAK_GBS * 1e9), 4),
                       "k_scan_extract_ms": round(tm2.get("k_scan_extract", 0.0), 4),
                       "outputs_equal_headline_run": same}
            del msa2, first
        cpu = None
        if world == 1 and a.cpu_baseline_mb > 0:
            cpu = cpu_baseline(ctx, torch, S, a.cpu_baseline_mb, a.variant_fraction, a.seed, l)
        line = {
            "metric": "msa2eds_input_MB_per_s", "value": round(value, 1), "unit": "MB/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "msa2eds %d-seq x %d-column synthetic alignment%s "
                                   "(genrandomeds-shaped, %.0f%% sites, one line per row)"
                                   % (S, a.cols if a.scaling == "strong" else a.cols * world,
                                      "" if world == 1 else " in %d column slabs of %d columns" % (world, L),
                                      a.variant_fraction * 100),
                       "rows": S, "cols_total": a.cols if a.scaling == "strong" else a.cols * world, "cols_per_gpu": L,
                       "context_len": l,
                       "input_bytes_per_gpu": n, "eds_bytes": out["E"], "seds_bytes": out["Q"],
                       "segments": info["n_segments"], "variant_cols": info["n_variant_cols"],
                       "slow_segments": info["n_slow_segments"],
                       "partition": "columns x %d" % world,
                       "stitch": (out.get("stitch") or {}).get("chains") if distributed else None},
            "frac_of_hbm_read_roofline": round(value * 1e6 / world / (HBM_PEAK_GBS * 1e9), 4),
            "roofline": roof, "cpu_baseline": cpu, "verify": verify, "aligned_rows": aligned, "kernel_ms": per_kernel,
        }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
