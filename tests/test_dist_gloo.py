"""CPU, world_size 2 over gloo: the data-parallel shard / gather path used by bench.py --gpus N."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "emma-x_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from emmax import dist as edist

    r, w, _ = edist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    total = 5                      # ragged: rank 0 gets 3 frames, rank 1 gets 2
    lo, hi = edist.shard_bounds(total, rank, world)
    counts = [edist.shard_bounds(total, i, world)[1] - edist.shard_bounds(total, i, world)[0] for i in range(world)]
    b, T = hi - lo, 6
    acts = torch.arange(lo, hi, dtype=torch.float32)[:, None] + torch.arange(7, dtype=torch.float32)[None] * 0.125
    ids = (torch.arange(lo, hi, dtype=torch.int32)[:, None] * 100 + torch.arange(T, dtype=torch.int32)[None])
    lens = torch.arange(lo, hi, dtype=torch.int32) + 1
    A, I, Ls = edist.gather_results(acts, ids, lens, counts)
    edist.barrier()
    mx = edist.max_over_ranks(float(rank + 1), "cpu")
    assert edist.collective_world_size("cpu") == world and edist.backend_name() == "gloo"
    q.put((rank, A.tolist(), I.tolist(), Ls.tolist(), mx))
    dist.destroy_process_group()


def test_gather_results_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp_A = [[i + j * 0.125 for j in range(7)] for i in range(5)]
    exp_I = [[i * 100 + t for t in range(6)] for i in range(5)]
    for rank, A, I, Ls, mx in outs:
        assert A == exp_A and I == exp_I and Ls == [1, 2, 3, 4, 5] and mx == 2.0


def test_shard_bounds_cover_everything():
    from emmax.dist import shard_bounds

    for n in (1, 7, 8, 64, 65):
        for w in (1, 2, 4, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_passthrough():
    from emmax.dist import gather_results

    a, i, l = torch.zeros(2, 7), torch.zeros(2, 3, dtype=torch.int32), torch.ones(2, dtype=torch.int32)
    A, I, L = gather_results(a, i, l)
    assert A is a and I is i and L is l


class _FakeModel:
    """Stands in for the GPU model in the CPU test of generate_actions_dp: row r of the global batch -> known outputs."""

    device = torch.device("cpu")

    def generate_actions_batch(self, frames, rows, max_new_tokens, stop_on_eos, tokenizer):
        import numpy as np

        b = frames.shape[0]
        tag = frames[:, 0, 0, 0].to(torch.int32)            # each frame carries its global row index in pixel (0,0,0)
        ids = tag[:, None] * 10 + torch.arange(max_new_tokens, dtype=torch.int32)[None]
        acts = np.stack([np.full(7, float(t)) for t in tag.tolist()]).astype(np.float32) if b else np.zeros((0, 7), np.float32)
        return acts, ids, tag + 1


def _dp_worker(rank, world, port, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "emma-x_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from emmax import dist as edist

    edist.init_from_env("gloo")
    total = 19                                              # rank 0: 10 rows (two sub-batches of 8 + 2), rank 1: 9
    frames = torch.zeros(total, 2, 2, 3, dtype=torch.uint8)
    frames[:, 0, 0, 0] = torch.arange(total, dtype=torch.uint8)
    A, I, L = edist.generate_actions_dp(_FakeModel(), frames, [[1]] * total, max_new_tokens=4)
    q.put((rank, A[:, 0].tolist(), I[:, 0].tolist(), L.tolist()))
    dist.destroy_process_group()


def test_generate_actions_dp_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, a0, i0, lens in outs:
        assert a0 == [float(i) for i in range(19)] and i0 == [10 * i for i in range(19)] and lens == [i + 1 for i in range(19)]
