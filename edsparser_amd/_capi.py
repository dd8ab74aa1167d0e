"""ctypes bindings of include/edsx.h."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STATUS = {0: "OK", 1: "FILE_NOT_FOUND", 2: "INVALID_FORMAT", 3: "INVALID_PARAMETER",
          4: "BUILD_FAILED", 5: "QUERY_FAILED", 99: "UNKNOWN_ERROR"}


class EdsxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (STATUS.get(code, code), msg))
        self.code = code
        self.message = msg


class _Buf(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("size", ctypes.c_size_t)]


class VcfStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in
                ("total_variants", "processed_variants", "skipped_malformed",
                 "skipped_unsupported_sv", "variant_groups")]


class EdsStatistics(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_uint64) for n in ("n_symbols", "n_chars", "n_strings", "num_degenerate_symbols",
                                                "total_change_size", "num_common_chars", "num_empty_strings",
                                                "min_context_length", "max_context_length", "num_context_blocks")] +
                [("avg_context_length", ctypes.c_double)] +
                [(n, ctypes.c_uint64) for n in ("has_sources", "num_paths", "max_paths_per_string", "total_paths")] +
                [("avg_paths_per_string", ctypes.c_double), ("is_leds", ctypes.c_int)])


class MsaInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in
                ("n_rows", "n_cols", "line_width", "n_variant_cols", "n_segments", "msa_bytes",
                 "n_slow_segments")]


class MsaAnchors(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("n_segments", "found", "first_seg", "last_seg", "last_col", "last_eds_bytes",
                                               "last_seds_bytes", "first_end", "first_eds_end", "first_seds_end")]


class MsaEdges(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in
                ("n_segments", "first_is_variant", "first_cols", "first_eds_bytes", "first_seds_bytes",
                 "last_is_variant", "last_cols", "last_eds_bytes", "last_seds_bytes")]


def lib_path():
    return os.environ.get("EDSX_LIB") or os.path.join(_HERE, "libedsx.so")


def load_library():
    """Load libedsx.so; fails loudly when the HIP extension has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch bundles its own HIP runtime (same SONAME libamdhip64.so.7): it must be the first one
    # mapped, so that libedsx.so binds to it instead of bringing a second runtime into the process
    import torch  # noqa: F401
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError("libedsx.so is missing: build it with `python -m edsparser_amd.build` "
                          "(there is no CPU fallback)")
    lib = ctypes.CDLL(path)
    P = ctypes.POINTER
    lib.edsx_version.restype = ctypes.c_char_p
    lib.edsx_ctx_create.argtypes = [ctypes.c_int, P(ctypes.c_void_p)]
    lib.edsx_ctx_destroy.argtypes = [ctypes.c_void_p]
    lib.edsx_last_error.argtypes = [ctypes.c_void_p]
    lib.edsx_last_error.restype = ctypes.c_char_p
    lib.edsx_buf_free.argtypes = [P(_Buf)]
    lib.edsx_msa_transform.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                                       P(_Buf), P(_Buf)]
    lib.edsx_msa_last_batches.argtypes = [ctypes.c_void_p]
    lib.edsx_msa_transform_batched.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int,
                                               P(_Buf), P(_Buf), P(ctypes.c_int)]
    lib.edsx_leds_merge.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                    ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int, P(_Buf), P(_Buf)]
    lib.edsx_vcf_transform.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                       ctypes.c_size_t, ctypes.c_uint32, P(_Buf), P(_Buf), P(VcfStats)]
    lib.edsx_eds_stats.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                   ctypes.c_uint32, P(EdsStatistics)]
    lib.edsx_genrandomeds.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_double, ctypes.c_uint32, ctypes.c_uint32,
                                      ctypes.c_uint32, ctypes.c_double, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64,
                                      P(_Buf), P(_Buf), P(ctypes.c_uint64)]
    lib.edsx_leds_tokenised_on_device.argtypes = [ctypes.c_void_p]
    lib.edsx_vcf_tokenised_on_device.argtypes = [ctypes.c_void_p]
    lib.edsx_leds_merge_range.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                          ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          P(_Buf), P(_Buf), P(ctypes.c_int), P(ctypes.c_int)]
    lib.edsx_vcf_index.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, P(_Buf), P(_Buf), P(_Buf), P(_Buf),
                                   P(VcfStats)]
    lib.edsx_vcf_sort_order.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.edsx_vcf_transform_range.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                             ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64, P(_Buf), P(_Buf),
                                             P(VcfStats)]
    lib.edsx_msa_plan_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                                         ctypes.c_void_p, P(ctypes.c_uint64), P(ctypes.c_uint64)]
    lib.edsx_msa_emit_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.edsx_msa_last_info.argtypes = [ctypes.c_void_p, P(MsaInfo)]
    lib.edsx_msa_edge_info.argtypes = [ctypes.c_void_p, P(MsaEdges)]
    lib.edsx_msa_anchor_info.argtypes = [ctypes.c_void_p, ctypes.c_uint64, P(MsaAnchors)]
    lib.edsx_msa_copy_columns.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
    lib.edsx_msa_locate_segment.argtypes = [ctypes.c_void_p, ctypes.c_uint64] + [P(ctypes.c_uint64)] * 4
    lib.edsx_set_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.edsx_get_timing.argtypes = [ctypes.c_void_p, P(ctypes.c_char_p), P(ctypes.c_float), P(ctypes.c_int),
                                    ctypes.c_int]
    lib.edsx_msa_synth_size.argtypes = [ctypes.c_uint32, ctypes.c_uint64]
    lib.edsx_msa_synth_size.restype = ctypes.c_size_t
    lib.edsx_msa_synth_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                                          ctypes.c_uint64, ctypes.c_uint64, ctypes.c_double, ctypes.c_uint64,
                                          ctypes.c_void_p, P(ctypes.c_size_t)]
    lib.edsx_multi_create.argtypes = [P(ctypes.c_int), ctypes.c_int, ctypes.c_int, P(ctypes.c_void_p)]
    lib.edsx_multi_destroy.argtypes = [ctypes.c_void_p]
    lib.edsx_multi_last_error.argtypes = [ctypes.c_void_p]
    lib.edsx_multi_last_error.restype = ctypes.c_char_p
    lib.edsx_msa_transform_multi.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, P(_Buf), P(_Buf)]
    lib.edsx_multi_last_partition.argtypes = [ctypes.c_void_p, P(ctypes.c_int), P(ctypes.c_int)]
    lib.edsx_genvcf.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64, P(_Buf), P(_Buf)]
    lib.edsx_msa_synth_size_aligned.argtypes = [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint32]
    lib.edsx_msa_synth_size_aligned.restype = ctypes.c_size_t
    lib.edsx_msa_synth_device_aligned.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                                                  ctypes.c_uint64, ctypes.c_uint64, ctypes.c_double, ctypes.c_uint64,
                                                  ctypes.c_uint32, ctypes.c_void_p, P(ctypes.c_size_t)]
    _LIB = lib
    return lib


class MultiGpu:
    """edsx_multi: MSA -> EDS over several GPUs from one process (C++ rank threads, RCCL or in-process exchange)."""

    def __init__(self, devices, use_rccl=True):
        self._lib = load_library()
        arr = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        rc = self._lib.edsx_multi_create(arr, len(devices), 1 if use_rccl else 0, ctypes.byref(h))
        if rc != 0:
            raise EdsxError(rc, "edsx_multi_create failed for devices %r (rccl=%r)" % (list(devices), use_rccl))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.edsx_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def msa_transform(self, msa, context_len=0):
        e, s = _Buf(), _Buf()
        if isinstance(msa, bytes):
            ptr, n, keep = msa, len(msa), msa
        else:
            import numpy as np
            keep = np.frombuffer(msa, dtype=np.uint8)
            ptr, n = ctypes.c_void_p(keep.ctypes.data), int(keep.size)
        rc = self._lib.edsx_msa_transform_multi(self._h, ptr, n, context_len, ctypes.byref(e), ctypes.byref(s))
        del keep
        if rc != 0:
            raise EdsxError(rc, self._lib.edsx_multi_last_error(self._h).decode(errors="replace"))
        out = []
        for b in (e, s):
            out.append(ctypes.string_at(b.data, b.size) if b.size else b"")
            self._lib.edsx_buf_free(ctypes.byref(b))
        return out[0], out[1]

    def last_partition(self):
        p, c = ctypes.c_int(), ctypes.c_int()
        self._lib.edsx_multi_last_partition(self._h, ctypes.byref(p), ctypes.byref(c))
        return bool(p.value), int(c.value)


def synth_size(n_rows, n_cols, row_align=0):
    if row_align > 1:
        return int(load_library().edsx_msa_synth_size_aligned(n_rows, n_cols, row_align))
    return int(load_library().edsx_msa_synth_size(n_rows, n_cols))


class Context:
    """One context per (thread, GPU); mirrors edsx_ctx_create/destroy."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = ctypes.c_void_p()
        rc = self._lib.edsx_ctx_create(device, ctypes.byref(h))
        if rc != 0:
            raise EdsxError(rc, "no usable gfx950 device %d (the engine has no CPU fallback)" % device)
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.edsx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EdsxError(rc, self._lib.edsx_last_error(self._h).decode(errors="replace"))

    def _take(self, b):
        data = ctypes.string_at(b.data, b.size) if b.size else b""
        self._lib.edsx_buf_free(ctypes.byref(b))
        return data

    # ---- host-buffer entry points
    def msa_transform(self, msa, context_len=0):
        """msa: bytes, or any object with the buffer protocol (a mapped file is handed over in place, not copied)."""
        e, s = _Buf(), _Buf()
        if isinstance(msa, bytes):
            ptr, n, keep = msa, len(msa), msa
        else:
            import numpy as np
            keep = np.frombuffer(msa, dtype=np.uint8)
            ptr, n = ctypes.c_void_p(keep.ctypes.data), int(keep.size)
        self._check(self._lib.edsx_msa_transform(self._h, ptr, n, context_len, ctypes.byref(e), ctypes.byref(s)))
        del keep
        return self._take(e), self._take(s)

    def msa_last_batches(self):
        return int(self._lib.edsx_msa_last_batches(self._h))

    def msa_transform_batched(self, msa, context_len=0, batches=2):
        """The same in `batches` column batches, one after the other on this GPU (bounded working set).  Returns
        (eds, seds, batches taken) - 1 when the input is not cut."""
        e, s = _Buf(), _Buf()
        used = ctypes.c_int(0)
        if isinstance(msa, bytes):
            ptr, n, keep = msa, len(msa), msa
        else:
            import numpy as np
            keep = np.frombuffer(msa, dtype=np.uint8)
            ptr, n = ctypes.c_void_p(keep.ctypes.data), int(keep.size)
        self._check(self._lib.edsx_msa_transform_batched(self._h, ptr, n, context_len, batches, ctypes.byref(e), ctypes.byref(s),
                                                         ctypes.byref(used)))
        del keep
        return self._take(e), self._take(s), used.value

    def leds_merge(self, eds, seds=None, context_len=1, compact=True):
        o, so = _Buf(), _Buf()
        eds = bytes(eds)
        sb = bytes(seds) if seds is not None else None
        self._check(self._lib.edsx_leds_merge(self._h, eds, len(eds), sb, len(sb) if sb is not None else 0,
                                              context_len, 1 if compact else 0, ctypes.byref(o), ctypes.byref(so)))
        return self._take(o), self._take(so)

    def eds_stats(self, eds, seds=None, context_len=0):
        """Statistics (EDS::Statistics) and is_leds(context_len) of an .eds (+ .seds) text as a dict."""
        st = EdsStatistics()
        self._check(self._lib.edsx_eds_stats(self._h, eds, len(eds), seds, len(seds) if seds is not None else 0,
                                             context_len, ctypes.byref(st)))
        return {n: getattr(st, n) for n, _ in EdsStatistics._fields_}

    def genrandomeds(self, total_bp, variability=0.10, min_alt=2, max_alt=4, var_len_max=10, snp_ratio=0.7,
                     alphabet="ACGT", min_context=0, seed=42):
        """genrandomeds-shaped (.eds, .seds, number of variant sites), generated on the GPU."""
        e, s = _Buf(), _Buf()
        n = ctypes.c_uint64(0)
        self._check(self._lib.edsx_genrandomeds(self._h, total_bp, variability, min_alt, max_alt, var_len_max, snp_ratio,
                                                alphabet.encode(), min_context, seed, ctypes.byref(e), ctypes.byref(s),
                                                ctypes.byref(n)))
        return self._take(e), self._take(s), n.value

    def vcf_transform(self, vcf, fasta, context_len=0):
        e, s, st = _Buf(), _Buf(), VcfStats()
        vcf, fasta = bytes(vcf), bytes(fasta)
        self._check(self._lib.edsx_vcf_transform(self._h, vcf, len(vcf), fasta, len(fasta), context_len,
                                                 ctypes.byref(e), ctypes.byref(s), ctypes.byref(st)))
        return self._take(e), self._take(s), {n: int(getattr(st, n)) for n, _ in VcfStats._fields_}

    def vcf_tokenised_on_device(self):
        return bool(self._lib.edsx_vcf_tokenised_on_device(self._h))

    def leds_tokenised_on_device(self):
        return bool(self._lib.edsx_leds_tokenised_on_device(self._h))

    # ---- symbol-range partition of the merge (multi-GPU, see multigpu.MergeSharder)
    def leds_merge_range(self, eds, seds=None, context_len=1, compact=True, head_sentinel=False, tail_sentinel=False):
        """-> (leds, seds_out, head_intact, tail_intact)"""
        o, so = _Buf(), _Buf()
        hi, ti = ctypes.c_int(), ctypes.c_int()
        eds = bytes(eds)
        sb = bytes(seds) if seds is not None else None
        self._check(self._lib.edsx_leds_merge_range(self._h, eds, len(eds), sb, len(sb) if sb is not None else 0,
                                                    context_len, 1 if compact else 0, 1 if head_sentinel else 0,
                                                    1 if tail_sentinel else 0, ctypes.byref(o), ctypes.byref(so),
                                                    ctypes.byref(hi), ctypes.byref(ti)))
        return self._take(o), self._take(so), bool(hi.value), bool(ti.value)

    # ---- position-range partition of the VCF path (multi-GPU, see multigpu.VcfSharder)
    def vcf_index(self, vcf):
        """(pos, reflen, line_off, line_len) as numpy uint64 arrays, file order, + counters."""
        import numpy as np
        bufs = [_Buf() for _ in range(4)]
        st = VcfStats()
        vcf = bytes(vcf)
        self._check(self._lib.edsx_vcf_index(self._h, vcf, len(vcf), *[ctypes.byref(b) for b in bufs], ctypes.byref(st)))
        arrs = [np.frombuffer(self._take(b), dtype=np.uint64) for b in bufs]
        return (*arrs, {n: int(getattr(st, n)) for n, _ in VcfStats._fields_})

    def vcf_sort_order(self, pos):
        """Permutation of the reference's std::sort for these positions (numpy uint64 -> uint32)."""
        import numpy as np
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        out = np.empty(len(pos), dtype=np.uint32)
        rc = self._lib.edsx_vcf_sort_order(pos.ctypes.data, len(pos), out.ctypes.data)
        if rc != 0:
            raise EdsxError(rc, "edsx_vcf_sort_order failed")
        return out

    def vcf_transform_range(self, vcf_lines, fasta, cur0=0, next_start=None):
        e, s, st = _Buf(), _Buf(), VcfStats()
        vcf_lines, fasta = bytes(vcf_lines), bytes(fasta)
        nxt = 0xFFFFFFFFFFFFFFFF if next_start is None else int(next_start)
        self._check(self._lib.edsx_vcf_transform_range(self._h, vcf_lines, len(vcf_lines), fasta, len(fasta), int(cur0),
                                                       nxt, ctypes.byref(e), ctypes.byref(s), ctypes.byref(st)))
        return self._take(e), self._take(s), {n: int(getattr(st, n)) for n, _ in VcfStats._fields_}

    # ---- device-resident entry points (pointers are ints: tensor.data_ptr())
    def msa_plan_device(self, d_msa, n, context_len=0, stream=0):
        E, Q = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self._lib.edsx_msa_plan_device(self._h, d_msa, n, context_len, stream,
                                                   ctypes.byref(E), ctypes.byref(Q)))
        return int(E.value), int(Q.value)

    def msa_emit_device(self, d_eds, d_seds, stream=0):
        self._check(self._lib.edsx_msa_emit_device(self._h, d_eds, d_seds, stream))

    def msa_info(self):
        info = MsaInfo()
        rc = self._lib.edsx_msa_last_info(self._h, ctypes.byref(info))
        if rc != 0:
            raise EdsxError(rc, "no planned alignment")
        return {n: int(getattr(info, n)) for n, _ in MsaInfo._fields_}

    def msa_edge_info(self):
        e = MsaEdges()
        self._check(self._lib.edsx_msa_edge_info(self._h, ctypes.byref(e)))
        return {n: int(getattr(e, n)) for n, _ in MsaEdges._fields_}

    def msa_anchor_info(self, min_cols):
        """First / last common segment of at least min_cols columns of the planned alignment (see edsx.h)."""
        a = MsaAnchors()
        self._check(self._lib.edsx_msa_anchor_info(self._h, min_cols, ctypes.byref(a)))
        return {n: int(getattr(a, n)) for n, _ in MsaAnchors._fields_}

    def msa_copy_columns(self, col0, ncols, n_rows):
        buf = ctypes.create_string_buffer(n_rows * ncols)
        self._check(self._lib.edsx_msa_copy_columns(self._h, col0, ncols, buf))
        return buf.raw

    def msa_locate_segment(self, col):
        """-> (segment index, start column, .eds offset, .seds offset) of the first segment starting at or after col"""
        v = [ctypes.c_uint64() for _ in range(4)]
        self._check(self._lib.edsx_msa_locate_segment(self._h, int(col), *[ctypes.byref(x) for x in v]))
        return tuple(int(x.value) for x in v)

    def set_timing(self, on):
        self._lib.edsx_set_timing(self._h, 1 if on else 0)

    def get_timing(self):
        cap = 64
        names = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_float * cap)()
        cnt = (ctypes.c_int * cap)()
        n = self._lib.edsx_get_timing(self._h, names, ms, cnt, cap)
        return [(names[i].decode(), float(ms[i]), int(cnt[i])) for i in range(n)]

    def genvcf(self, ref_len, n_records, n_samples=8, seed=42):
        """Synthetic (vcf, fasta) bytes of BASELINE configs[3]'s shape, generated on the device."""
        v, f = _Buf(), _Buf()
        self._check(self._lib.edsx_genvcf(self._h, ref_len, n_records, n_samples, seed, ctypes.byref(v), ctypes.byref(f)))
        return self._take(v), self._take(f)

    def msa_synth_device(self, d_out, capacity, n_rows, n_cols, col0=0, variant_fraction=0.05, seed=42,
                         stream=0, row_align=0):
        w = ctypes.c_size_t()
        if row_align > 1:
            self._check(self._lib.edsx_msa_synth_device_aligned(self._h, d_out, capacity, n_rows, col0, n_cols,
                                                                variant_fraction, seed, row_align, stream, ctypes.byref(w)))
        else:
            self._check(self._lib.edsx_msa_synth_device(self._h, d_out, capacity, n_rows, col0, n_cols,
                                                        variant_fraction, seed, stream, ctypes.byref(w)))
        return int(w.value)
