"""Run by test_msa_gpu.py::test_weak_signature_build in a process of its own, with EDSX_LIB pointing at
libedsx_weaksig.so (msa_device.hip compiled with -DEDSX_TEST_WEAK_SIG: one bit of row signature).  Different rows
of a wide variant segment then collide all the time; the byte-for-byte verification behind the signatures must
still give the oracle's bytes.  Prints "weaksig ok <cases> <slow segments>"."""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import edsparser_amd            # noqa: E402
import oracle_lib as o          # noqa: E402
from msa_cases import campaign_msa, wide_msa   # noqa: E402

assert "weaksig" in edsparser_amd.lib_path(), edsparser_amd.lib_path()
ctx = edsparser_amd.Context(0)
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 77)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
slow = 0
for it in range(n):
    msa = wide_msa(rng, nul=(it % 5 == 4)) if it % 4 else campaign_msa(rng)[0]
    for l in (0, rng.choice([1, 3, 9])):
        try:
            want = o.msa(msa, l)
        except o.OracleError as ex:
            want = ("ERR", str(ex))
        try:
            got = ctx.msa_transform(msa, l)
        except edsparser_amd.EdsxError as ex:
            got = ("ERR", ex.message)
        if got != want:
            open(os.path.join(os.path.dirname(HERE), "gpurun_out", "weaksig_fail_%d_%d.msa" % (it, l)), "wb").write(msa)
            print("weaksig MISMATCH case", it, "l", l)
            sys.exit(1)
        if l == 0 and not isinstance(got[0], str):
            slow += ctx.msa_info()["n_slow_segments"]
print("weaksig ok", n, slow)
