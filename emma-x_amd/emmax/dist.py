"""
dist.py -- data-parallel sharding of camera frames over the GPUs of one node (one process per GPU).

The reference's inference is strictly single-process / single-GPU (experiments/robot/openvla_utils.py:21,57); this is
the extension of BASELINE config 3: every frame is an independent unit, each rank holds a full bf16 weight replica and a
private KV cache, and the only exchange is ONE all_gather per batch of the results
{actions f32 [b,7], token ids i32 [b,T], lengths i32 [b]} (~17 KB per rank) -- RCCL over xGMI on the GPU box
(`backend="nccl"` is RCCL on ROCm), gloo in the CPU tests.  No tensor parallelism, no per-token communication.
"""

from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def _singleton() -> bool:
    """EMMAX_DIST_SINGLETON=1: a one-rank launch still initialises the process group and runs every collective for real -- the
    way the RCCL code path (communicator set-up, all_gather / all_reduce on device buffers) is exercised on a one-GPU box."""
    return os.environ.get("EMMAX_DIST_SINGLETON", "0") == "1"


def _collectives_on() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or _singleton())


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's env; initialises the default group when WORLD_SIZE > 1 (or EMMAX_DIST_SINGLETON=1)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or (_singleton() and "MASTER_PORT" in os.environ)) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            # EMMAX_DIST_BACKEND=gloo lets the multi-process control flow be exercised on a single-GPU box
            backend = os.environ.get("EMMAX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of `n_items` for `rank`: the first (n % world) ranks get one extra item."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_results(actions: torch.Tensor, ids: torch.Tensor, lens: torch.Tensor, counts: Optional[Sequence[int]] = None):
    """One collective: every rank contributes its shard, every rank receives the whole batch in global order.

    actions f32 [b,7], ids i32 [b,T], lens i32 [b] -> ([B,7], [B,T], [B]).  Shards may be ragged in b (`counts` =
    per-rank shard sizes; default: all equal): rows are packed into one fixed-size i32 buffer per rank so a single
    all_gather moves everything."""
    if not _collectives_on():
        return actions, ids, lens
    world = dist.get_world_size()
    b, T = ids.shape
    counts = list(counts) if counts is not None else [b] * world
    bmax = max(counts)
    width = 7 + T + 1
    pack = torch.zeros(bmax, width, dtype=torch.int32, device=ids.device)
    pack[:b, :7] = actions.to(torch.float32).contiguous().view(torch.int32)
    pack[:b, 7:7 + T] = ids.to(torch.int32)
    pack[:b, 7 + T] = lens.to(torch.int32)
    out = torch.empty(world * bmax, width, dtype=torch.int32, device=ids.device)
    if dist.get_backend() == "gloo" and pack.is_cuda:   # test hook only: gloo moves the 2 KB/row through the host
        out_h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(out_h, pack.cpu())
        out.copy_(out_h)
    else:
        dist.all_gather_into_tensor(out, pack)
    out = out.view(world, bmax, width)
    rows = torch.cat([out[r, : counts[r]] for r in range(world)], dim=0)
    return rows[:, :7].contiguous().view(torch.float32), rows[:, 7:7 + T].contiguous(), rows[:, 7 + T].contiguous()


def barrier() -> None:
    if _collectives_on():
        dist.barrier()


def max_over_ranks(x: float, device) -> float:
    if not _collectives_on():
        return x
    t = torch.tensor([x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


MAX_ROWS = 64   # rows of one decode batch (EMMAX_MAX_DECODE_BATCH, emma-x_amd/csrc/kernels.h)


def generate_actions_dp(model, frames_u8: torch.Tensor, prompt_rows, max_new_tokens: int = 512, stop_on_eos: bool = True,
                        tokenizer=None):
    """Data-parallel `generate_actions_batch` (BASELINE config 3): every rank passes the SAME global batch (frames uint8
    [B,H,W,3], B prompts); rank r computes its contiguous shard on its own GPU (sub-batches of <= 64 rows) and one
    all_gather returns (actions f32 [B,7], ids i32 [B,T], lens i32 [B]) in global order on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    Btot = frames_u8.shape[0]
    lo, hi = shard_bounds(Btot, rank, world)
    counts = [shard_bounds(Btot, r, world)[1] - shard_bounds(Btot, r, world)[0] for r in range(world)]
    dev = model.device
    acts_l, ids_l, lens_l = [], [], []
    eng = getattr(model, "engine", None)
    step = min(MAX_ROWS, eng.max_decode_batch()) if eng is not None and hasattr(eng, "max_decode_batch") else 8
    for s0 in range(lo, hi, step):
        s1 = min(hi, s0 + step)
        a, i, n = model.generate_actions_batch(frames_u8[s0:s1].to(dev).contiguous(), prompt_rows[s0:s1], max_new_tokens,
                                               stop_on_eos, tokenizer)
        acts_l.append(torch.from_numpy(a).to(dev))
        ids_l.append(i)
        lens_l.append(n)
    if acts_l:
        acts, ids, lens = torch.cat(acts_l), torch.cat(ids_l), torch.cat(lens_l)
    else:   # more ranks than frames
        acts = torch.zeros(0, 7, dtype=torch.float32, device=dev)
        ids = torch.zeros(0, max_new_tokens, dtype=torch.int32, device=dev)
        lens = torch.zeros(0, dtype=torch.int32, device=dev)
    return gather_results(acts, ids, lens, counts)


def backend_name() -> str:
    return dist.get_backend() if dist.is_initialized() else "none"


def collective_world_size(device) -> int:
    """Ranks that took part in an ACTUAL collective (all_reduce of ones): what `bench.py` reports as `rccl_ranks`."""
    if not _collectives_on():
        return 1
    t = torch.ones(1, dtype=torch.int32, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
