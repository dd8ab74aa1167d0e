"""CPU: libedsx.so loads and exports every symbol include/edsx.h declares (no compute calls)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    import edsparser_amd.build as b
    lib = b.build()
    hdr = open(os.path.join(ROOT, "include", "edsx.h")).read()
    declared = set(re.findall(r"\b(edsx_[a-z_]+)\s*\(", hdr))
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (edsx_[a-z_]+)", out))
    assert declared and declared <= exported, sorted(declared - exported)


def test_library_loads_without_gpu_and_fails_loudly():
    import edsparser_amd
    lib = edsparser_amd.load_library()
    assert lib.edsx_version().startswith(b"edsx")
    import torch
    if not torch.cuda.is_available():
        import pytest
        with pytest.raises(edsparser_amd.EdsxError):
            edsparser_amd.Context(0)
