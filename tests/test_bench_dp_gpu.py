"""The driver's multi-GPU launch line, on ONE GPU: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`
with both ranks forced onto device 0 and gloo standing in for RCCL (EMMAX_FORCE_DEVICE / EMMAX_DIST_BACKEND test hooks).
Checks the control flow of the N > 1 path end to end -- rendezvous, per-rank shards, the one result gather, max-over-ranks
timing, rank-0 JSON line with `rccl_ranks` / `gather_ms` -- that SURVEY.md 8e asks the bench line to carry."""

import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_tiny_gloo(device):
    env = dict(os.environ, EMMAX_FORCE_DEVICE="0", EMMAX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--tiny",
           "--prompt-tokens", "24", "--new-tokens", "12"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout            # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["dist_backend"] == "gloo"
    assert d["config"]["batch_per_gpu"] == 8 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1
    assert abs(d["value"] - 16 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]      # whole-job rate: all ranks' frames / max-over-ranks time
    assert d["gather_ms"] > 0 and 0 < d["gather_share"] < 0.5
    assert "cpu_baseline" not in d                 # rank 0 at N = 1 only
    assert d["roofline"]["bound"] == "hbm"
