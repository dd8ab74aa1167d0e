"""Build libedsx.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

One object per source under edsparser_amd/build/ (compiled in parallel, only when stale), then one link.
`libedsx_weaksig.so` is a TEST-ONLY variant: msa_device.hip compiled with -DEDSX_TEST_WEAK_SIG (row signatures
of the multi-column grouping masked to one bit, so that different rows collide all the time and the byte-for-byte
verification behind the signatures is what keeps the output right); tests/test_msa_gpu.py runs it against the oracle.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libedsx.so")
LIB_WEAKSIG = os.path.join(HERE, "libedsx_weaksig.so")
SOURCES = ["msa_device.hip", "merge_device.hip", "vcf_device.hip", "synth.hip", "stats_device.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hs.append(os.path.join(HERE, "..", "include", "edsx.h"))
    return hs


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, obj, extra, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + extra + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)


def _link(objs, lib, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    sources = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    objs = []
    for s in sources:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            jobs.append((src, obj, []))
    weak_obj = os.path.join(OBJ, "msa_device_weaksig.o")
    msa_src = os.path.join(CSRC, "msa_device.hip")
    if force or _newer(weak_obj, [msa_src] + hdrs):
        jobs.append((msa_src, weak_obj, ["-DEDSX_TEST_WEAK_SIG"]))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(lambda j: _compile(j[0], j[1], j[2], verbose), jobs))
    if force or _newer(LIB, objs):
        _link(objs, LIB, verbose)
    weak_objs = [weak_obj if o.endswith("msa_device.o") else o for o in objs]
    if force or _newer(LIB_WEAKSIG, weak_objs):
        _link(weak_objs, LIB_WEAKSIG, verbose)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
