#!/usr/bin/env python3
"""Third set of reference-generated fixtures (round 3): the text of EDS::print_statistics and EDS::print
(/root/reference/src/cpp/lib/formats/eds.cpp:528-598) for a few inputs, produced by the REAL reference library
(oracle/_ref/libedsref.so, build container only) and committed as data for tests/cpp/test_container.cpp.

File format of print_cases.txt, per case:  <eds>\n<seds or ->\n<n>\n<n bytes of print_statistics>\n<m>\n<m bytes of print>\n

    python tests/golden/make_golden3.py            # regenerate
    python tests/golden/make_golden3.py --check    # re-run the reference, compare with the committed file
"""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as o  # noqa: E402

CASES = [
    (b"{ACGTAC}{A,ACA,}{CGTTTTT}{,T}{GG}{C,G}{TTTTTTTTTT}", b"{0}{1,3}{2}{4}{0}{1,2}{3,4}{0}{1,2,3}{4}{0}"),
    (b"{ACGTAC}{A,ACA,}{CGTTTTT}{,T}{GG}{C,G}{TTTTTTTTTT}", None),
    (b"ACGT{A,C}GG", None),
    (b"{A,C}{G,T}", b"{1}{2}{1,2}{3}"),
    (b"{ACGT}", None),
    (b"{,A}{,}{TTT}{G,GG,GGG}", None),
]


def ref_print(eds, seds, which):
    lib = o._load_ref()
    out = ctypes.c_char_p()
    n = ctypes.c_size_t()
    err = ctypes.create_string_buffer(512)
    lib.ref_eds_print.restype = ctypes.c_int
    rc = lib.ref_eds_print(eds, ctypes.c_size_t(len(eds)), seds, ctypes.c_size_t(len(seds) if seds else 0), ctypes.c_int(which),
                           ctypes.byref(out), ctypes.byref(n), err, ctypes.c_size_t(512))
    if rc:
        raise RuntimeError(err.value.decode())
    b = ctypes.string_at(out, n.value)
    lib.ref_free.argtypes = [ctypes.c_void_p]
    lib.ref_free(out)
    return b


def render():
    parts = []
    for eds, seds in CASES:
        st, pr = ref_print(eds, seds, 0), ref_print(eds, seds, 1)
        parts += [eds, b"\n", seds or b"-", b"\n", str(len(st)).encode(), b"\n", st, b"\n", str(len(pr)).encode(), b"\n", pr, b"\n"]
    return b"".join(parts)


if __name__ == "__main__":
    path = os.path.join(HERE, "print_cases.txt")
    data = render()
    if "--check" in sys.argv:
        assert open(path, "rb").read() == data, "print_cases.txt differs from the reference's output"
        print("print_cases.txt: reference agrees (%d cases)" % len(CASES))
    else:
        open(path, "wb").write(data)
        print("wrote %s (%d bytes)" % (path, len(data)))
