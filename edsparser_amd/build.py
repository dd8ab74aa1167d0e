"""Build libedsx.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

One object per source under edsparser_amd/build/ (compiled in parallel, only when stale), then one link.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libedsx.so")
SOURCES = ["msa_device.hip", "msa_scan.hip", "merge_device.hip", "vcf_device.hip", "synth.hip", "genrandom.hip", "genvcf.hip", "multi_gpu.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# per-source flags.  msa_scan.hip: machine scheduler off - the column scan's loads stay in source order (bench shape: scan
# 26.5 instead of 26.9 ms; the same flag on msa_device.hip costs its emitters 0.2 ms: alternating A/B runs, EXPERIMENTS.md)
EXTRA_FLAGS = {"msa_scan.hip": ["-mllvm", "-enable-misched=false"]}
# A second library for the tests only: the merge's final-text kernel with a 256-entry LDS stack and a 2-entry register stack
# in its serial walk, so that ordinary inputs take the overflow -> serial walk -> HBM spill-stack path that otherwise needs
# merge trees deeper than 96 (tests/test_merge_gpu.py::test_deep_tree_fallback_paths).  Never loaded by the product.
TEST_LIB = os.path.join(HERE, "libedsx_smallstacks.so")
TEST_FLAGS = {"merge_device.hip": ["-DEDSX_EXPERIMENTS", "-DEDSX_FW_STACK=256", "-DEDSX_FIN_STACK=2"]}


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
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]   # RCCL: the multi-GPU boundary stitch
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
            jobs.append((src, obj, EXTRA_FLAGS.get(s, [])))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(lambda j: _compile(j[0], j[1], j[2], verbose), jobs))
    if force or _newer(LIB, objs):
        _link(objs, LIB, verbose)
    tobjs = list(objs)
    for name, extra in TEST_FLAGS.items():
        src, obj = os.path.join(CSRC, name), os.path.join(OBJ, name.replace(".hip", ".smallstacks.o"))
        tobjs[tobjs.index(os.path.join(OBJ, name.replace(".hip", ".o")))] = obj
        if force or _newer(obj, [src] + hdrs):
            _compile(src, obj, extra, verbose)
    if force or _newer(TEST_LIB, tobjs):
        _link(tobjs, TEST_LIB, verbose)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
