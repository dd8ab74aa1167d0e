"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when built, the real reference
library (oracle/_ref/libedsref.so).  TEST INFRASTRUCTURE: imported by tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() only — never by edsparser_amd/."""
import ctypes
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class VcfStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in
                ("total_variants", "processed_variants", "skipped_malformed",
                 "skipped_unsupported_sv", "variant_groups")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class EdsStatistics(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_uint64) for n in ("n_symbols", "n_chars", "n_strings", "num_degenerate_symbols",
                                                "total_change_size", "num_common_chars", "num_empty_strings",
                                                "min_context_length", "max_context_length", "num_context_blocks")] +
                [("avg_context_length", ctypes.c_double)] +
                [(n, ctypes.c_uint64) for n in ("has_sources", "num_paths", "max_paths_per_string", "total_paths")] +
                [("avg_paths_per_string", ctypes.c_double), ("is_leds", ctypes.c_int)])

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def _stats_call(fn, eds, seds, l):
    e, en = _buf(eds)
    s, sn = _buf(seds)
    st = EdsStatistics()
    err = ctypes.create_string_buffer(512)
    rc = fn(e, ctypes.c_size_t(en), s, ctypes.c_size_t(sn), ctypes.c_uint32(l), ctypes.byref(st), err, ctypes.c_size_t(512))
    if rc != 0:
        raise OracleError(rc, err.value.decode(errors="replace"))
    return st.as_dict()


def eds_stats(eds, seds=None, l=0):
    """EDS::Statistics + is_leds(l) by the oracle restatement."""
    return _stats_call(_load().oracle_eds_stats, eds, seds, l)


def ref_eds_stats(eds, seds=None, l=0):
    """... by the real reference (build container only); num_context_blocks / total_paths are not reported (0)."""
    return _stats_call(_load_ref().ref_eds_stats, eds, seds, l)


def build_oracle():
    """Compile oracle/ (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-s", "-C", _ORACLE_DIR], check=True)


_lib = None
_ref = None


def _load():
    global _lib
    if _lib is None:
        so = os.path.join(_ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so):
            build_oracle()
        _lib = ctypes.CDLL(so)
        _lib.oracle_free.argtypes = [ctypes.c_void_p]
    return _lib


def have_ref():
    return os.path.exists(os.path.join(_ORACLE_DIR, "_ref", "libedsref.so"))


def _load_ref():
    global _ref
    if _ref is None:
        _ref = ctypes.CDLL(os.path.join(_ORACLE_DIR, "_ref", "libedsref.so"))
        _ref.ref_free.argtypes = [ctypes.c_void_p]
    return _ref


def _buf(b):
    if b is None:
        return None, 0
    b = bytes(b)
    return ctypes.create_string_buffer(b, len(b)) if len(b) else ctypes.create_string_buffer(1), len(b)


def _take(free, p, n):
    data = ctypes.string_at(p, n.value) if n.value else b""
    free(p)
    return data


def _call(fn, free, args_pre, n_out=2, extra=()):
    outs = [(ctypes.c_void_p(), ctypes.c_size_t()) for _ in range(n_out)]
    err = ctypes.create_string_buffer(512)
    flat = []
    for p, n in outs:
        flat += [ctypes.byref(p), ctypes.byref(n)]
    rc = fn(*args_pre, *flat, *extra, err, ctypes.c_size_t(512))
    if rc != 0:
        raise OracleError(rc, err.value.decode(errors="replace"))
    return [_take(free, p, n) for p, n in outs]


def msa(file_bytes, l=0):
    lib = _load()
    b, n = _buf(file_bytes)
    return tuple(_call(lib.oracle_msa, lib.oracle_free, [b, ctypes.c_size_t(n), ctypes.c_uint32(l)]))


def msa_slab(file_bytes, c0, c1):
    lib = _load()
    b, n = _buf(file_bytes)
    return tuple(_call(lib.oracle_msa_slab, lib.oracle_free,
                       [b, ctypes.c_size_t(n), ctypes.c_uint64(c0), ctypes.c_uint64(c1)]))


def merge(eds, seds=None, l=1, compact=True):
    lib = _load()
    e, en = _buf(eds)
    s, sn = _buf(seds)
    return tuple(_call(lib.oracle_merge, lib.oracle_free,
                       [e, ctypes.c_size_t(en), s, ctypes.c_size_t(sn), ctypes.c_uint32(l),
                        ctypes.c_int(1 if compact else 0)]))


def merge_range(eds, seds=None, l=1, compact=True, head_sentinel=False, tail_sentinel=False):
    """-> (leds, seds_out, head_intact, tail_intact)"""
    lib = _load()
    e, en = _buf(eds)
    s, sn = _buf(seds)
    hi, ti = ctypes.c_int(), ctypes.c_int()
    a, b = _call(lib.oracle_merge_range, lib.oracle_free,
                 [e, ctypes.c_size_t(en), s, ctypes.c_size_t(sn), ctypes.c_uint32(l), ctypes.c_int(1 if compact else 0),
                  ctypes.c_int(1 if head_sentinel else 0), ctypes.c_int(1 if tail_sentinel else 0)],
                 extra=(ctypes.byref(hi), ctypes.byref(ti)))
    return a, b, bool(hi.value), bool(ti.value)


def vcf(vcf_bytes, fasta_bytes, l=0):
    lib = _load()
    v, vn = _buf(vcf_bytes)
    f, fn = _buf(fasta_bytes)
    st = VcfStats()
    e, s = _call(lib.oracle_vcf, lib.oracle_free,
                 [v, ctypes.c_size_t(vn), f, ctypes.c_size_t(fn), ctypes.c_uint32(l)],
                 extra=(ctypes.byref(st),))
    return e, s, st.as_dict()


def vcf_range(vcf_lines, fasta_bytes, cur0=0, next_start=None):
    lib = _load()
    v, vn = _buf(vcf_lines)
    f, fn = _buf(fasta_bytes)
    st = VcfStats()
    nxt = 0xFFFFFFFFFFFFFFFF if next_start is None else int(next_start)
    e, s = _call(lib.oracle_vcf_range, lib.oracle_free,
                 [v, ctypes.c_size_t(vn), f, ctypes.c_size_t(fn), ctypes.c_uint64(int(cur0)), ctypes.c_uint64(nxt)],
                 extra=(ctypes.byref(st),))
    return e, s, st.as_dict()


def vcf_index(vcf_bytes):
    import numpy as np
    lib = _load()
    v, vn = _buf(vcf_bytes)
    cap = bytes(vcf_bytes).count(b"\n") + 1
    arrs = [np.zeros(cap, dtype=np.uint64) for _ in range(4)]
    n = ctypes.c_size_t()
    st = VcfStats()
    rc = lib.oracle_vcf_index(v, ctypes.c_size_t(vn), *[ctypes.c_void_p(a.ctypes.data) for a in arrs],
                              ctypes.c_size_t(cap), ctypes.byref(n), ctypes.byref(st))
    if rc != 0:
        raise OracleError(rc, "oracle_vcf_index failed")
    return (*[a[:n.value].copy() for a in arrs], st.as_dict())


def vcf_sort_order(pos):
    import numpy as np
    lib = _load()
    pos = np.ascontiguousarray(pos, dtype=np.uint64)
    out = np.empty(len(pos), dtype=np.uint32)
    lib.oracle_vcf_sort_order(ctypes.c_void_p(pos.ctypes.data), ctypes.c_size_t(len(pos)), ctypes.c_void_p(out.ctypes.data))
    return out


# ---- the real reference (build container only; merge + VCF paths) ----

def ref_merge(eds, seds=None, l=1, compact=True, threads=1):
    lib = _load_ref()
    e, en = _buf(eds)
    s, sn = _buf(seds)
    return tuple(_call(lib.ref_merge, lib.ref_free,
                       [e, ctypes.c_size_t(en), s, ctypes.c_size_t(sn), ctypes.c_uint32(l),
                        ctypes.c_int(1 if compact else 0), ctypes.c_int(threads)]))


def ref_vcf(vcf_bytes, fasta_bytes, l=0):
    lib = _load_ref()
    v, vn = _buf(vcf_bytes)
    f, fn = _buf(fasta_bytes)
    st = VcfStats()
    e, s = _call(lib.ref_vcf, lib.ref_free,
                 [v, ctypes.c_size_t(vn), f, ctypes.c_size_t(fn), ctypes.c_uint32(l)],
                 extra=(ctypes.byref(st),))
    return e, s, st.as_dict()
