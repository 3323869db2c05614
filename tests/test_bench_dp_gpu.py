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
    _check_live_scaling_fields(d)


def _check_live_scaling_fields(d):
    """An N > 1 line carries its own one-GPU baseline, measured in the same job, and nothing read from a committed file."""
    one = d["single_gpu_same_workload"]
    assert one["source"].startswith("live") and one["value"] > 0
    assert abs(d["scaling_efficiency"] - d["value"] / (d["n_gpus"] * one["value"])) < 1e-3
    assert d["roofline"]["traffic"] is None and d["roofline"]["traffic_source"] is None
    assert "--batch-per-gpu %d" % d["config"]["batch_per_gpu"] in d["cmd"] and "--gpus %d" % d["n_gpus"] in d["cmd"]


def test_bench_self_launch_two_ranks_tiny_gloo(device):
    """`python bench.py --gpus 2` with no launcher around it re-executes itself under torch.distributed.run (the N = 1 driver line
    is a plain `python bench.py --gpus 1`; the same spelling at N > 1 must not die on WORLD_SIZE)."""
    env = dict(os.environ, EMMAX_FORCE_DEVICE="0", EMMAX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--tiny",
           "--prompt-tokens", "24", "--new-tokens", "12", "--batch-per-gpu", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["config"]["batch_per_gpu"] == 3 and d["config"]["global_batch"] == 6
    _check_live_scaling_fields(d)
    # --scale-baseline replaces the live measurement
    out = subprocess.run(cmd + ["--scale-baseline", "123.5"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["single_gpu_same_workload"]["value"] == 123.5 and "command line" in d["single_gpu_same_workload"]["source"]


def test_bench_one_rank_rccl_singleton(device):
    """RCCL itself on the box: the driver's launch line with ONE rank and EMMAX_DIST_SINGLETON=1 -- the `nccl` (= RCCL) process group is
    created on the device, and the result all_gather, the barrier and the max-over-ranks / rank-count all_reduces run through it on
    device buffers.  (Two ranks cannot share one GPU under RCCL; with one rank the communicator set-up, the packing of
    {actions, ids, lens} into the int32 gather buffer and its unpacking are the code the 8-GPU run executes.)"""
    env = dict(os.environ, EMMAX_DIST_SINGLETON="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("EMMAX_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--tiny",
           "--prompt-tokens", "24", "--new-tokens", "12", "--batch-per-gpu", "3", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["dist_backend"] == "nccl" and d["rccl_ranks"] == 1 and d["n_gpus"] == 1
    assert d["gather_ms"] > 0            # the all_gather ran (a world-1 short cut would report no gather time)


def test_gather_results_through_rccl_is_the_identity_at_world_one(device):
    """`gather_results` on device buffers through the nccl backend, one rank: pack -> all_gather_into_tensor -> unpack returns the
    inputs bit for bit (fp32 actions travel as int32 words)."""
    code = r"""
import os, sys, torch
sys.path[:0] = [%r, %r]
os.environ.update(EMMAX_DIST_SINGLETON="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=%r)
from emmax import dist as ed
rank, world, local = ed.init_from_env()
assert ed.backend_name() == "nccl", ed.backend_name()
g = torch.Generator().manual_seed(5)
a = torch.randn(5, 7, generator=g).cuda(); ids = torch.randint(0, 32000, (5, 19), generator=g, dtype=torch.int32).cuda()
lens = torch.randint(1, 19, (5,), generator=g, dtype=torch.int32).cuda()
A, I, N = ed.gather_results(a, ids, lens)
assert A.data_ptr() != a.data_ptr()          # went through the gather buffer
assert torch.equal(A.view(torch.int32), a.view(torch.int32)) and torch.equal(I, ids) and torch.equal(N, lens)
assert ed.collective_world_size("cuda:0") == 1 and ed.max_over_ranks(1.5, "cuda:0") == 1.5
ed.barrier()
torch.distributed.destroy_process_group()
print("rccl-singleton-ok")
""" % (ROOT, os.path.join(ROOT, "emma-x_amd"), str(_free_port()))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT)
    assert out.returncode == 0 and "rccl-singleton-ok" in out.stdout, out.stderr[-3000:]
