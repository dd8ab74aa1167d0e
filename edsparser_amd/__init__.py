"""edsparser_amd — MI355X-native EDS transformation engine.

Python here is plumbing only: a ctypes view of the C ABI in include/edsx.h (libedsx.so, built by
edsparser_amd/build.py) used by the tests and bench.py.  The product is the HIP library and the
C++ `edsparser::` host shims in edsparser_amd/host/.  There is no CPU fallback: importing works
anywhere, but creating a Context without a gfx950 GPU raises.
"""
from ._capi import Context, EdsxError, MultiGpu, lib_path, load_library, synth_size  # noqa: F401
