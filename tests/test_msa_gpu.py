"""GPU parity tests for the MSA -> EDS / l-EDS path: the HIP pipeline, called through the C ABI
(edsx_msa_transform / edsx_msa_plan_device + edsx_msa_emit_device), must be byte-identical to
the CPU oracle and to the reference's golden vectors."""
import json
import os
import random

import pytest

import oracle_lib as o
from conftest import GOLDEN
from msa_cases import campaign_msa, random_msa, wide_msa

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import edsparser_amd
    c = edsparser_amd.Context(0)
    yield c
    c.close()


def _kat():
    return json.load(open(os.path.join(GOLDEN, "kat_msa.json")))["cases"]


@pytest.mark.parametrize("case", _kat(), ids=lambda c: c["name"])
def test_reference_vectors(ctx, case):
    msa = case["msa"].encode() if "msa" in case else open(os.path.join(GOLDEN, case["msa_file"]), "rb").read()
    eds, seds = ctx.msa_transform(msa, case["l"])
    assert eds.decode() == case["eds"]
    assert seds.decode() == case["seds"]


@pytest.mark.parametrize("l", [0, 1, 2, 3, 5, 8, 32])
def test_random_small_alignments(ctx, l):
    rng = random.Random(1000 + l)
    for i in range(60):
        msa = random_msa(rng, trailing_newline=(i % 3 != 0))
        want = o.msa(msa, l)
        got = ctx.msa_transform(msa, l)
        assert got == want, (l, i, msa)


def test_many_rows_and_wide_rows(ctx):
    rng = random.Random(7)
    for S, L, lw in [(100, 300, None), (1000, 70, 70), (1030, 40, 40), (2, 5000, 5000), (300, 1000, 60),
                     (4097, 33, 33), (5000, 20, 20)]:
        msa = random_msa(rng, S=S, L=L, lw=lw)
        for l in (0, 4):
            assert ctx.msa_transform(msa, l) == o.msa(msa, l), (S, L, lw, l)


def test_format_errors(ctx):
    import edsparser_amd
    for bad in [b"", b"ACGT\n", b">a\nACGT\n", b">a\nACGT\n>b\nAC\n", b">a\nACGT\n>b\nACGTA\n",
                b">a\nAC\nGT\n>b\nACG\nT\n"]:
        with pytest.raises(edsparser_amd.EdsxError) as ei:
            ctx.msa_transform(bad, 0)
        assert ei.value.code == 2, bad


def test_row_limit_is_a_build_limit_not_a_format_error(ctx):
    """The row limit of the build is the width of a sequence id in the .seds text (seven digits + ',' = one 8-byte
    token): 10 000 000 sequences are reported as EDSX_ERR_BUILD_FAILED (4) with their own text, not as a format error."""
    import edsparser_amd
    big = b">\nA\n" * 10_000_000
    with pytest.raises(edsparser_amd.EdsxError) as ei:
        ctx.msa_transform(big, 0)
    assert ei.value.code == 4 and "more sequences than this build supports" in ei.value.message


@pytest.mark.parametrize("S,L", [(8193, 400), (12000, 300), (70001, 60)])
def test_more_rows_than_the_lds_holds(ctx, S, L):
    """More than 8192 sequences: the generic kernels keep their row tables in HBM scratch (32-bit group ids), the column
    scan reads row starts from HBM and writes variant bytes straight to vc, the row table is grown past its first 65536
    entries.  One-line and wrapped rows, l = 0 and an l-EDS, against the oracle."""
    rng = random.Random(77 + S)
    for lw in (None, 61):
        msa = random_msa(rng, S=S, L=L, lw=lw, p_var=0.06)
        for l in (0, 3):
            got = ctx.msa_transform(msa, l)
            want = o.msa(msa, l)
            assert got == want, (S, lw, l)


def test_more_than_65535_strings_in_one_segment(ctx):
    """70 000 rows of ten random letters: one variant segment with more distinct strings than a 16-bit group id counts."""
    import numpy as np
    rng = np.random.default_rng(5)
    S, L = 70_000, 10
    cells = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(S, L))]
    msa = b"".join(b">r%d\n" % i + cells[i].tobytes() + b"\n" for i in range(S))
    assert len({cells[i].tobytes() for i in range(S)}) > 65535
    assert ctx.msa_transform(msa, 0) == o.msa(msa, 0)


def test_trailing_blank_lines_tolerated(ctx):
    msa = b">a\nACGT\n>b\nACGA\n\n\n"
    assert ctx.msa_transform(msa, 0) == o.msa(msa, 0)


@pytest.mark.parametrize("S,L,l", [(48, 20000, 0), (48, 20000, 10), (64, 200000, 0), (1000, 30000, 0),
                                    (1000, 30000, 6), (7, 100003, 0),
                                    # l-EDS at the bench's row count: mixed segments through the heavy grouping kernel
                                    # (<= 64 columns), > 64 strings / > 64 columns through the generic kernels (staged
                                    # mixed segments, four-wave .seds walk, grouping cache)
                                    (1000, 60000, 3), (1000, 60000, 10), (1000, 40000, 32), (300, 50000, 10), (64, 100000, 10),
                                    # more than 1024 rows: the generic kernels take every variant segment
                                    (2000, 20000, 0), (2000, 20000, 5), (3000, 6000, 0)])
def test_synthetic_device_path(ctx, S, L, l):
    """genrandomeds-shaped alignment generated in HBM, device-resident plan/emit vs the oracle."""
    import torch
    import edsparser_amd
    n = edsparser_amd.synth_size(S, L)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    ctx.msa_synth_device(buf.data_ptr(), n, S, L, seed=42 + S)
    E, Q = ctx.msa_plan_device(buf.data_ptr(), n, l)
    d_eds = torch.empty(E + 16, dtype=torch.uint8, device="cuda:0")
    d_seds = torch.empty(Q + 16, dtype=torch.uint8, device="cuda:0")
    ctx.msa_emit_device(d_eds.data_ptr(), d_seds.data_ptr())
    torch.cuda.synchronize()
    host = bytes(buf.cpu().numpy())
    assert host.count(b">") == S
    oe, os_ = o.msa(host, l)
    assert bytes(d_eds[:E].cpu().numpy()) == oe
    assert bytes(d_seds[:Q].cpu().numpy()) == os_
    info = ctx.msa_info()
    assert info["n_rows"] == S and info["n_cols"] == L


def test_synthetic_slabs_match_whole(ctx):
    """Column slabs generated separately are the same bytes as the columns of the whole."""
    import torch
    import edsparser_amd
    S, L = 9, 4000
    n = edsparser_amd.synth_size(S, L)
    whole = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    ctx.msa_synth_device(whole.data_ptr(), n, S, L, seed=5)
    rows = [r.split(b"\n")[1] for r in bytes(whole.cpu().numpy()).split(b">")[1:]]
    for c0, c1 in [(0, 1000), (1000, 2500), (2500, 4000)]:
        m = edsparser_amd.synth_size(S, c1 - c0)
        part = torch.empty(m, dtype=torch.uint8, device="cuda:0")
        ctx.msa_synth_device(part.data_ptr(), m, S, c1 - c0, col0=c0, seed=5)
        prow = [r.split(b"\n")[1] for r in bytes(part.cpu().numpy()).split(b">")[1:]]
        assert prow == [r[c0:c1] for r in rows]


def _msa_with_headers(rng, headers, L, lw=None, trailing="\n", inject=None):
    ref = [rng.choice("ACGT") for _ in range(L)]
    out = []
    for i, h in enumerate(headers):
        row = list(ref)
        for c in range(0, L, 37):
            if rng.random() < 0.5:
                row[c] = rng.choice("ACGT-")
        if inject and i == inject[0]:
            row[inject[1]] = inject[2]
        s = "".join(row)
        out.append(h)
        w = lw or L
        out.extend(s[k:k + w] for k in range(0, L, w))
    return ("\n".join(out) + trailing).encode()


def test_row_index_speculation_and_fallback(ctx):
    """The row starts come from a speculative parallel search that is validated on the device, with the
    serial header chain as fallback: equal-length headers (speculation holds), headers whose lengths
    drift away from the first one (windows miss -> fallback), a '>' inside a data line, blank lines at
    the end, no final newline.  (A '>' at the START of a wrapped data line is a header line to the
    reference, i.e. a ragged alignment: rejected input)"""
    rng = random.Random(5)
    S, L = 40, 6000
    cases = [
        _msa_with_headers(rng, [">seq%03d" % i for i in range(S)], L),
        _msa_with_headers(rng, [">seq%03d" % i for i in range(S)], L, lw=60),
        _msa_with_headers(rng, [">s" + "x" * (3 * i) for i in range(S)], L),                 # drifting header lengths
        _msa_with_headers(rng, [">s%d" % (10 ** (i % 7)) for i in range(S)], L, lw=100),
        _msa_with_headers(rng, [">seq%03d" % i for i in range(S)], L, lw=60, inject=(7, 121, ">")),      # '>' inside a line
        _msa_with_headers(rng, [">seq%03d" % i for i in range(S)], L, trailing="\n\n\n"),
        _msa_with_headers(rng, [">seq%03d" % i for i in range(S)], L, trailing=""),
        _msa_with_headers(rng, [">r%d" % i for i in range(300)], 9000, lw=70),
    ]
    for i, msa in enumerate(cases):
        for l in (0, 5):
            assert ctx.msa_transform(msa, l) == o.msa(msa, l), (i, l)


def test_baseline_config_64x10mb_bit_exact(ctx):
    """BASELINE.json configs[1] at full size: 64 sequences x 10 Mb (640 MB), device-resident path,
    every output byte against the oracle (about 3 s of CPU work), plus l-EDS with l = 10."""
    import hashlib
    import torch
    import edsparser_amd
    S, L = 64, 10_000_000
    n = edsparser_amd.synth_size(S, L)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    ctx.msa_synth_device(buf.data_ptr(), n, S, L, variant_fraction=0.05, seed=42)
    torch.cuda.synchronize()
    host = bytes(buf.cpu().numpy())
    for l in (0, 10):
        E, Q = ctx.msa_plan_device(buf.data_ptr(), n, l)
        d_eds = torch.empty(E + 16, dtype=torch.uint8, device="cuda:0")
        d_seds = torch.empty(Q + 16, dtype=torch.uint8, device="cuda:0")
        ctx.msa_emit_device(d_eds.data_ptr(), d_seds.data_ptr())
        torch.cuda.synchronize()
        oe, os_ = o.msa(host, l)
        assert (E, Q) == (len(oe), len(os_)), l
        assert hashlib.sha256(d_eds[:E].cpu().numpy().tobytes()).digest() == hashlib.sha256(oe).digest(), l
        assert hashlib.sha256(d_seds[:Q].cpu().numpy().tobytes()).digest() == hashlib.sha256(os_).digest(), l


@pytest.mark.parametrize("S", [2, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513,
                               959, 960, 961, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 8192])
def test_row_count_boundaries(ctx, S):
    """Row counts around every layout switch: rows per lane 1/2/4/8/16 (vc pitch, K1 lane rows), the
    S <= 256 kernel instantiations, the last partial block of 64 rows, the fast/generic limit at 1024, the generic
    kernels' hash-table limit at 4096 rows and the LDS limit of their row tables at 8192."""
    rng = random.Random(1000 + S)
    for lw in (None, 61):
        msa = random_msa(rng, S=S, L=2500, lw=lw, p_var=0.06)
        for l in (0, 3):
            assert ctx.msa_transform(msa, l) == o.msa(msa, l), (S, lw, l)


def test_output_independent_of_slot_allocation_order(ctx):
    """Variant-column slots are handed out by an atomic counter, in a different order on every run;
    the text must not depend on it: two runs over the same 5 GB alignment (bench shape: 1000 rows,
    5 % sites) and a run with l = 7 twice give identical bytes."""
    import torch
    import edsparser_amd
    S, L = 1000, 5_000_000
    n = edsparser_amd.synth_size(S, L)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    ctx.msa_synth_device(buf.data_ptr(), n, S, L, variant_fraction=0.05, seed=11)
    for l in (0, 7):
        outs = []
        for _ in range(2):
            E, Q = ctx.msa_plan_device(buf.data_ptr(), n, l)
            d_eds = torch.empty(E + 16, dtype=torch.uint8, device="cuda:0")
            d_seds = torch.empty(Q + 16, dtype=torch.uint8, device="cuda:0")
            ctx.msa_emit_device(d_eds.data_ptr(), d_seds.data_ptr())
            torch.cuda.synchronize()
            outs.append((E, Q, d_eds[:E].clone(), d_seds[:Q].clone()))
        assert outs[0][:2] == outs[1][:2]
        assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][3], outs[1][3]), l
        # brace structure: every segment contributes exactly one '{' ... '}' pair per string set
        assert int((outs[0][2] == ord("{")).sum()) == int((outs[0][2] == ord("}")).sum())
        assert int((outs[0][3] == ord("{")).sum()) == int((outs[0][3] == ord("}")).sum())


def test_randomized_campaign(ctx):
    """250 alignments of the randomized campaign (tests/msa_cases.py: campaign_msa), EDS and one random
    context length each; 9000 further cases of the same generator were run once on the MI355X box
    (seeds 2-7) without a mismatch."""
    import edsparser_amd
    rng = random.Random(1)
    for it in range(250):
        msa, desc = campaign_msa(rng)
        for l in (0, rng.choice([1, 2, 5, 9, 33])):
            try:
                want = o.msa(msa, l)
            except o.OracleError as ex:
                want = ("ERR", str(ex))
            try:
                got = ctx.msa_transform(msa, l)
            except edsparser_amd.EdsxError as ex:
                got = ("ERR", ex.message)
            assert got == want, (it, l, desc)


@pytest.mark.parametrize("L", [1_200_003, 1_200_064, 1_200_777])
def test_host_buffer_path_large_outputs_equal_device_path(ctx, L):
    """edsx_msa_transform (host bytes in, host bytes out) with a .seds of ≈40 MB: the text comes back through the
    pinned-chunk download into a huge-page-advised buffer and must be the bytes the device-resident path holds in HBM
    (torch's own copy), for output sizes with different remainders."""
    import hashlib
    import torch
    import edsparser_amd
    S = 200
    n = edsparser_amd.synth_size(S, L)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    ctx.msa_synth_device(buf.data_ptr(), n, S, L, seed=L)
    E, Q = ctx.msa_plan_device(buf.data_ptr(), n, 0)
    d_eds = torch.empty(E + 16, dtype=torch.uint8, device="cuda:0")
    d_seds = torch.empty(Q + 16, dtype=torch.uint8, device="cuda:0")
    ctx.msa_emit_device(d_eds.data_ptr(), d_seds.data_ptr())
    torch.cuda.synchronize()
    assert Q >= 16 << 20
    want_e, want_s = d_eds[:E].cpu().numpy().tobytes(), d_seds[:Q].cpu().numpy().tobytes()
    eds, seds = ctx.msa_transform(bytes(buf.cpu().numpy()), 0)
    assert (len(eds), len(seds)) == (E, Q)
    assert hashlib.sha256(eds).digest() == hashlib.sha256(want_e).digest()
    assert hashlib.sha256(seds).digest() == hashlib.sha256(want_s).digest()


def test_two_contexts_download_side_by_side(ctx):
    """Two contexts of one process, each driven by its own host thread, both with outputs that take the pinned-chunk
    download (one staging set per device, no process-wide state): same bytes as a single call."""
    import hashlib
    import threading
    import torch
    import edsparser_amd
    S, L = 200, 1_200_003
    n = edsparser_amd.synth_size(S, L)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    ctx.msa_synth_device(buf.data_ptr(), n, S, L, seed=7)
    torch.cuda.synchronize()
    host = bytes(buf.cpu().numpy())
    want = ctx.msa_transform(host, 0)
    assert len(want[1]) >= 16 << 20
    others = [edsparser_amd.Context(0), edsparser_amd.Context(0)]
    got = [None, None]

    def work(i):
        got[i] = others[i].msa_transform(host, 0)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for g in got:
        assert g is not None and hashlib.sha256(g[0]).digest() == hashlib.sha256(want[0]).digest()
        assert hashlib.sha256(g[1]).digest() == hashlib.sha256(want[1]).digest()
    for c in others:
        c.close()


def _same_as_oracle(ctx, msa, l):
    import edsparser_amd
    try:
        want = o.msa(msa, l)
    except o.OracleError as ex:
        want = ("ERR", str(ex))
    try:
        got = ctx.msa_transform(msa, l)
    except edsparser_amd.EdsxError as ex:
        got = ("ERR", ex.message)
    return got == want


def test_wide_segments_are_grouped_exactly(ctx):
    """Variant segments of 2..70 columns over DNA, lower-case and protein alphabets, rows that differ only in
    where their gaps sit (so that different raw rows spell one string), strings of any length: the
    column-by-column refinement of the wave-per-segment kernels is exact for every byte value, and raw groups
    that spell the same gap-stripped string are joined (msa_transforms.cpp:262-293)."""
    rng = random.Random(20)
    for it in range(160):
        msa = wide_msa(rng)
        for l in (0, rng.choice([1, 4, 12])):
            assert _same_as_oracle(ctx, msa, l), (it, l)


def test_nul_bytes_end_a_rows_string(ctx):
    """The reference stops reading a row's segment bytes at a NUL (msa_transforms.cpp:282).  NUL bytes inside
    variant columns (single- and multi-column segments, any row incl. the first) and columns that are NUL in
    every row must give the oracle's bytes."""
    rng = random.Random(21)
    for it in range(120):
        msa = wide_msa(rng, nul=True)
        for l in (0, rng.choice([1, 4, 12])):
            assert _same_as_oracle(ctx, msa, l), (it, l)
    for it in range(120):                                  # isolated single-column sites with NULs
        msa = bytearray(random_msa(rng, S=rng.choice([2, 5, 70, 300, 1000]), L=rng.randint(5, 400), lw=None))
        body = [i for i, b in enumerate(msa) if b in b"ACGTacgtN-"]
        for i in rng.sample(body, min(len(body), rng.randint(1, 6))):
            msa[i] = 0
        for l in (0, 3):
            assert _same_as_oracle(ctx, bytes(msa), l), (it, l)


def test_full_size_config4_sampled_against_oracle(ctx):
    """BASELINE configs[4] at FULL size (1000 sequences x 10^8 columns, 100 GB resident in HBM, .seds 17 GB):
    40+ column windows spread over the whole width -- including the last columns and .seds offsets far beyond
    4 GiB -- are regenerated with the counter-based generator, run through the oracle and compared with the
    corresponding byte ranges of the device outputs; brace structure and the .seds size rule over the whole
    outputs (tests/fullsize_verify.py).  This is what pins the 64-bit offset paths of the scans and emitters."""
    import torch
    import edsparser_amd
    from fullsize_verify import verify_windows
    free, total = torch.cuda.mem_get_info()
    S, L = 1000, 100_000_000
    n = edsparser_amd.synth_size(S, L)
    if free < n + 60 * (1 << 30):
        pytest.skip("needs about 160 GB of free HBM")
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    ctx.msa_synth_device(buf.data_ptr(), n, S, L, variant_fraction=0.05, seed=42)
    E, Q = ctx.msa_plan_device(buf.data_ptr(), n, 0)
    assert Q > (1 << 32)
    d_eds = torch.empty(E + 16, dtype=torch.uint8, device="cuda:0")
    d_seds = torch.empty(Q + 16, dtype=torch.uint8, device="cuda:0")
    ctx.msa_emit_device(d_eds.data_ptr(), d_seds.data_ptr())
    torch.cuda.synchronize()
    res = verify_windows(ctx, torch, S, L, 42, 0.05, d_eds, d_seds, E, Q, nwin=40, width=12000)
    assert res["windows"] >= 32 and res["max_seds_offset_compared"] == Q and res["size_rule"] == "ok", res
    del buf, d_eds, d_seds
    torch.cuda.empty_cache()


@pytest.mark.parametrize("S,L,vf", [(50, 20000, 0.3), (300, 40000, 0.5), (1000, 20000, 0.9)])
def test_dense_variant_columns_grow_the_column_store(ctx, S, L, vf):
    """More variant columns than the first guess of the variant-column store holds (1/8 of the columns): the
    plan overflows, the host grows the store and replans.  Every kernel of the first attempt must stay inside its
    buffers (the slots past the capacity exist only as numbers)."""
    import torch
    import edsparser_amd
    n = edsparser_amd.synth_size(S, L)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    fresh = edsparser_amd.Context(0)                       # a context whose buffers have not grown yet
    fresh.msa_synth_device(buf.data_ptr(), n, S, L, variant_fraction=vf, seed=7)
    E, Q = fresh.msa_plan_device(buf.data_ptr(), n, 0)
    d_eds = torch.empty(E + 16, dtype=torch.uint8, device="cuda:0")
    d_seds = torch.empty(Q + 16, dtype=torch.uint8, device="cuda:0")
    fresh.msa_emit_device(d_eds.data_ptr(), d_seds.data_ptr())
    torch.cuda.synchronize()
    oe, os_ = o.msa(bytes(buf.cpu().numpy()), 0)
    assert bytes(d_eds[:E].cpu().numpy()) == oe
    assert bytes(d_seds[:Q].cpu().numpy()) == os_
    fresh.close()


def test_host_call_uploads_large_alignments_as_an_aligned_row_image(ctx):
    """edsx_msa_transform re-lays plain uniform alignments of 1 MB and more as a row image with every row on a multiple
    of 128 bytes (csrc/multi_gpu.hip upload_row_image); smaller ones and odd files are copied as they are.  Same bytes
    out: one-line and wrapped rows, header lengths that change from row to row, with and without the final newline,
    context lengths, and a ragged file (whose error the transform words)."""
    import edsparser_amd
    rng = random.Random(4242)
    for it, (S, L, lw, nl) in enumerate([(5, 300_000, None, True), (9, 150_000, 60, True), (3, 400_000, 70_001, False),
                                          (12, 100_000, None, False), (2, 600_000, 7, True)]):
        msa = random_msa(rng, S=S, L=L, lw=lw, trailing_newline=nl, p_var=0.03)
        assert len(msa) >= 1 << 20
        lines = msa.split(b"\n")                       # headers of different lengths: the rows are not equally spaced
        k = 0
        for i, ln in enumerate(lines):
            if ln.startswith(b">"):
                lines[i] = ln + b"x" * (k % 5)
                k += 1
        msa = b"\n".join(lines)
        for l in ((0, 4) if it < 2 else (0,)):
            assert ctx.msa_transform(msa, l) == o.msa(msa, l), (it, l)
    bad = bytearray(random_msa(rng, S=4, L=400_000))
    del bad[-5:-3]                                     # the last row is two columns short
    with pytest.raises(edsparser_amd.EdsxError) as ei:
        ctx.msa_transform(bytes(bad), 0)
    assert "Invalid MSA" in str(ei.value)


@pytest.mark.parametrize("K", [2, 3, 7])
def test_column_batches_equal_the_single_call(ctx, K):
    """edsx_msa_transform_batched: K column batches one after the other through one pipeline, stitched like the slabs of
    the multi-GPU path (boundary columns straight from the host image).  l = 0 and l-EDS, one-line and wrapped rows."""
    import edsparser_amd
    rng = random.Random(900 + K)
    cut = whole = 0
    for i in range(40):
        lw = rng.choice([None, None, 7, 60])
        l = rng.choice([0, 0, 1, 3, 8])
        msa = random_msa(rng, S=rng.randint(2, 9), L=rng.randint(40 * K, 1200), lw=lw, trailing_newline=rng.random() < 0.7,
                         p_var=rng.choice([0.01, 0.05, 0.2]))
        e, s, used = ctx.msa_transform_batched(msa, l, K)
        assert (e, s) == o.msa(msa, l), (i, l, msa)
        assert used in (1, K)
        cut += used == K
        whole += used == 1
        if used == K:                                       # the pipeline holds the plan of the last batch only: no "last info"
            with pytest.raises(edsparser_amd.EdsxError):
                ctx.msa_info()
        else:
            assert ctx.msa_info()["n_cols"] > 0
    assert cut >= 25
    # not an alignment the geometry walk accepts: one piece, the transform words the error
    import edsparser_amd
    with pytest.raises(edsparser_amd.EdsxError) as ei:
        ctx.msa_transform_batched(b">a\nACGT\n>b\nACG\n", 0, K)
    assert ei.value.code == 2


def test_alignment_that_does_not_fit_in_one_piece_is_batched(ctx):
    """edsx_msa_transform falls back to column batches when the device runs out of memory: most of the free HBM is taken
    away, then an alignment whose tables need more than what is left (but whose batches do not) must still come out
    byte-equal to the run with the memory available."""
    import torch
    import edsparser_amd
    S, L = 64, 24_000_000
    n = edsparser_amd.synth_size(S, L)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    ctx.msa_synth_device(buf.data_ptr(), n, S, L, seed=5)
    torch.cuda.synchronize()
    host = bytes(buf.cpu().numpy())
    del buf
    want = ctx.msa_transform(host, 0)
    c2 = edsparser_amd.Context(0)                                   # a fresh context: nothing cached from the run above
    torch.cuda.empty_cache()
    free, _total = torch.cuda.mem_get_info()
    keep_free = 3 << 30                                             # the whole run needs ~5 GB (image 1.5 GB + 84 B per column of tables)
    hog = torch.empty(max(0, free - keep_free), dtype=torch.uint8, device="cuda:0")
    try:
        got = c2.msa_transform(host, 0)
        batches = c2.msa_last_batches()
    finally:
        del hog
        torch.cuda.empty_cache()
        c2.close()
    assert got == want
    assert batches >= 2


def test_row_loop_kernels_and_their_fallbacks(ctx):
    """More than 1024 rows: a wave walks the rows of a variant segment 64 at a time (msa_rowloop_kernels.hpp) when it has
    at most sixteen pure variant columns and at most 64 strings; a wider run, a segment with more strings, a NUL byte and
    (l > 0) mixed segments go to the generic kernels.  Constructed alignment: 40 narrow sites, runs of 12 and of 20 columns, one
    site with 100 different letters/strings, one site with a NUL - the routing is checked through n_slow_segments."""
    import numpy as np
    rng = np.random.default_rng(17)
    for S in (1025, 1500, 4100, 9000):
        L = 600
        ref = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=L)]
        rows = np.tile(ref, (S, 1))
        sites = list(range(10, 410, 10))                                     # 40 narrow sites (1..2 columns with gaps: at most 25 strings)
        for c in sites:
            w = int(rng.integers(1, 3))
            for j in range(w):
                alt = np.frombuffer(b"ACGT-", dtype=np.uint8)[rng.integers(0, 5, size=S)]
                pick = rng.random(S) < 0.4
                pick[0] = False
                rows[pick, c + j] = alt[pick]
        wide = slice(450, 470)                                                # a run of 20 variant columns
        rows[1::2, wide] = np.frombuffer(b"T" * 20, dtype=np.uint8)
        rows[0, wide] = np.frombuffer(b"G" * 20, dtype=np.uint8)
        rows[2, wide] = np.frombuffer(b"G" * 20, dtype=np.uint8)
        rows[3, 430:442] = np.frombuffer(b"CCCC--AAAAAA", dtype=np.uint8)      # a run of 12 columns: still the row-loop kernels
        rows[4, 430:442] = np.frombuffer(b"AAAAAATTTTTT", dtype=np.uint8)
        many = 500                                                            # two columns, 100 distinct strings
        rows[0, many:many + 2] = np.frombuffer(b"AA", dtype=np.uint8)
        for r in range(1, S):
            v = r % 100
            rows[r, many] = 65 + v // 10
            rows[r, many + 1] = 97 + v % 10
        rows[5, 550] = 0                                                      # a NUL byte (ends the row's string)
        rows[6, 550] = ord("T") if ref[550] != ord("T") else ord("A")
        msa = b"".join(b">s%d\n" % i + rows[i].tobytes() + b"\n" for i in range(S))
        assert _same_as_oracle(ctx, msa, 0), S
        slow = ctx.msa_info()["n_slow_segments"]
        assert slow == 3, (S, slow)                                           # the wide run, the 100-string site, the NUL site
        assert _same_as_oracle(ctx, msa, 4), S                                # l-EDS: mixed segments take the generic kernels
        assert ctx.msa_info()["n_slow_segments"] >= 3


def test_row_chain_with_headers_of_every_length(ctx):
    """The row index when headers vary in length (the speculative parallel index does not apply): the chain requests the
    windows of the next eight rows on the prediction that their headers are as long as the last one.  Header lengths that
    drift slowly, jump, alternate, exceed the 64-byte window, and the `>s<idx>` style whose length grows with the digits;
    2 .. 5000 rows; with and without a final newline."""
    rng = random.Random(64)
    styles = [lambda i: "s%d" % i, lambda i: "seq", lambda i: "x" * (1 + i % 3), lambda i: "n" * rng.randint(0, 40),
              lambda i: "L" * (rng.choice([3, 3, 3, 70, 130])), lambda i: "d" * (5 + i // 50), lambda i: "a" if i % 2 else "b" * 33]
    for S in (2, 9, 65, 700, 5000):
        for st in styles:
            L = rng.choice([1, 7, 40, 300])
            ref = "".join(rng.choice("ACGT") for _ in range(L))
            rows = []
            for i in range(S):
                r = list(ref)
                if i and rng.random() < 0.3:
                    r[rng.randrange(L)] = rng.choice("ACGT-")
                rows.append("".join(r))
            text = "".join(">%s\n%s\n" % (st(i), rows[i]) for i in range(S))
            if rng.random() < 0.3:
                text = text[:-1]
            msa = text.encode()
            for l in (0, 2):
                assert ctx.msa_transform(msa, l) == o.msa(msa, l), (S, L, l)
