"""Sampled parity check of a device-resident MSA -> EDS run that is too large for the oracle as a whole
(BASELINE configs[4]: 1000 rows x 10^8 columns, 100 GB in, 17 GB of .seds).  TEST INFRASTRUCTURE: used by
tests/test_msa_gpu.py and by bench.py's --verify leg (outside the timed region); calls the oracle.

With context length 0 the segments are the maximal runs of common / variant columns, so the text of the
segments that start inside a column window [a, b) cut at segment starts is exactly what the transform gives
for the columns [a, b) alone.  The synthetic generator is counter-based (every byte depends on seed, global
column and row only), so any window can be regenerated, run through the CPU oracle and compared with the
corresponding byte ranges of the device outputs (edsx_msa_locate_segment gives the offsets).  Windows are
spread over the whole width, always include the first and the last columns, and so cover .seds offsets far
beyond 4 GiB.  On top of that the whole outputs are checked for their brace structure and the size rule
|.seds| = 3 * (#common) + (#variant) * tokens(S) + (#strings of variant segments).
"""
import oracle_lib as o


def token_total(S):
    return sum(len(str(r)) + 1 for r in range(1, S + 1))


def _count_byte(torch, t, n, ch, chunk=1 << 30):
    tot = 0
    for off in range(0, n, chunk):
        tot += int((t[off:min(n, off + chunk)] == ch).sum().item())
    return tot


def verify_windows(ctx, torch, S, L, seed, vfrac, d_eds, d_seds, E, Q, nwin=40, width=12000, col0=0, structure=True):
    """Returns a dict describing what was compared; raises AssertionError on the first difference."""
    import edsparser_amd
    info = ctx.msa_info()
    nseg = info["n_segments"]
    assert info["n_cols"] == L and info["n_rows"] == S
    width = min(width, L)
    starts = sorted(set([0, L - width] + [int(i * (L - width) / max(1, nwin - 1)) for i in range(nwin)]))
    windows, eds_bytes, seds_bytes, max_seds_off, max_eds_off = 0, 0, 0, 0, 0
    for w0 in starts:
        w1 = min(L, w0 + width)
        sa, ca, ea, qa = ctx.msa_locate_segment(w0)
        sb, cb, eb, qb = ctx.msa_locate_segment(w1) if w1 < L else (nseg, L, E, Q)
        if cb <= ca:
            continue
        n = edsparser_amd.synth_size(S, cb - ca)
        buf = torch.empty(n, dtype=torch.uint8, device=d_eds.device)
        ctx.msa_synth_device(buf.data_ptr(), n, S, cb - ca, col0=col0 + ca, variant_fraction=vfrac, seed=seed)
        torch.cuda.synchronize()
        host = bytes(buf.cpu().numpy())
        del buf
        oe, os_ = o.msa(host, 0)
        ge = d_eds[ea:eb].cpu().numpy().tobytes()
        gs = d_seds[qa:qb].cpu().numpy().tobytes()
        assert len(ge) == len(oe) and ge == oe, ("eds differs", w0, ca, cb, ea, eb, len(oe))
        assert len(gs) == len(os_) and gs == os_, ("seds differs", w0, ca, cb, qa, qb, len(os_))
        windows += 1
        eds_bytes += len(oe)
        seds_bytes += len(os_)
        max_seds_off = max(max_seds_off, qb)
        max_eds_off = max(max_eds_off, eb)
    res = {"windows": windows, "window_cols": width, "eds_bytes_compared": eds_bytes,
           "seds_bytes_compared": seds_bytes, "max_seds_offset_compared": max_seds_off,
           "max_eds_offset_compared": max_eds_off, "last_window_reaches_end": max_seds_off == Q and max_eds_off == E}
    assert res["last_window_reaches_end"], res
    if structure:
        edges = ctx.msa_edge_info()
        nvar = (nseg + 1) // 2 if edges["first_is_variant"] else nseg // 2
        ncommon = nseg - nvar
        eo, ec = _count_byte(torch, d_eds, E, ord("{")), _count_byte(torch, d_eds, E, ord("}"))
        qo, qc = _count_byte(torch, d_seds, Q, ord("{")), _count_byte(torch, d_seds, Q, ord("}"))
        assert eo == ec == nseg, (eo, ec, nseg)
        assert qo == qc, (qo, qc)
        strings = qo - ncommon                                   # id lists of the variant segments
        assert Q == 3 * ncommon + nvar * token_total(S) + strings, (Q, ncommon, nvar, strings)
        assert int(d_eds[0].item()) == ord("{") and int(d_eds[E - 1].item()) == ord("}")
        assert int(d_seds[0].item()) == ord("{") and int(d_seds[Q - 1].item()) == ord("}")
        res.update({"segments": nseg, "variant_segments": nvar, "variant_strings": strings, "size_rule": "ok"})
    return res
