#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point edsx_msa_transform (never bench.py's value):
host FASTA image -> H2D -> kernels -> D2H of .eds/.seds.  Usage: python3 profiles/host_path.py [cols]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, edsparser_amd

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
S = 1000
ctx = edsparser_amd.Context(0)
n = edsparser_amd.synth_size(S, cols)
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
ctx.msa_synth_device(buf.data_ptr(), n, S, cols, variant_fraction=0.05, seed=42)
torch.cuda.synchronize()
host = bytes(buf.cpu().numpy())
del buf
import ctypes
from edsparser_amd import _capi
def c_call():
    e, s = _capi._Buf(), _capi._Buf()
    t0 = time.perf_counter()
    rc = ctx._lib.edsx_msa_transform(ctx._h, host, len(host), 0, ctypes.byref(e), ctypes.byref(s))
    dt = time.perf_counter() - t0
    assert rc == 0
    sizes = (e.size, s.size)
    ctx._lib.edsx_buf_free(ctypes.byref(e)); ctx._lib.edsx_buf_free(ctypes.byref(s))
    return dt, sizes
c_call()                                                   # warm-up (allocations, pinned ring)
best = min(c_call()[0] for _ in range(3))
dt, (E, Q) = c_call()
print("host path (C ABI call only): %d input bytes, %.1f MB eds, %.1f MB seds, best %.3f s -> %.2f GB/s input (PCIe-inclusive)"
      % (n, E / 1e6, Q / 1e6, best, n / best / 1e9))
