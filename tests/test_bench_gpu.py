"""bench.py as the driver starts it for N > 1 (torch.distributed.run, one rank per GPU), rehearsed on the one-GPU box:
two ranks sharing the card over gloo (the multi-rank logic: slabs, stitch, reductions, one JSON line from rank 0), and the
production backend (nccl = RCCL) with the single rank the box allows (--force-dist: process group on the device, the
stitcher's tensor collectives and the all_reduces on device tensors)."""
import json
import os
import subprocess
import sys

import pytest

from test_vcf_shard_cpu import _free_port

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "1", "--cols", "4000000", "--cpu-baseline-mb", "0", "--verify", "0"]


def _run(nproc, extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + COMMON + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_ranks_share_the_gpu_over_gloo():
    j = _run(2, ["--backend", "gloo", "--share-gpu"])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["metric"] == "msa2eds_input_MB_per_s"
    assert j["config"]["cols_total"] == 4000000 and j["config"]["cols_per_gpu"] == 2000000
    assert j["value"] > 0 and j["steps"] == 2 and j["warmup"] == 1
    assert j["config"]["stitch"] == 1                         # one boundary between the two slabs: one chain


def test_rccl_single_rank_takes_the_multi_rank_path():
    j = _run(1, ["--backend", "nccl", "--force-dist"])
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["config"]["stitch"] is not None
