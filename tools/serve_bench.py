"""Continuous batching vs static batching on one MI355X (not the headline bench).

Emma-X-7B shapes, margin-boosted synthetic weights whose greedy chain is known a priori: request i emits k_i - 1 ordinary
ids, the 29871 prefix, 8 action ids and EOS, so the natural lengths k_i + 9 are spread like real reasoning chains of
different depth.  Every request = one 224x224 frame + a 32-token prompt.
  static      batches of `slots` requests through generate(); a batch ends when its longest row reaches EOS
  continuous  SlotScheduler: a finished slot is refilled at once
  + early     the same with the stop rule (29871 + 8 ids): EOS is never decoded
Agreement of the emitted ids between the modes is reported per request (see the comment in main)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
import numpy as np
import torch

from emmax.config import EmmaXConfig
from emmax.modeling import EmmaXForActionPrediction
from emmax.serving import Request, SlotScheduler
from emmax.weights import planted_chain, planted_start_token


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--requests", type=int, default=32)
    ap.add_argument("--slots", type=int, default=8)
    ap.add_argument("--poll", type=int, default=4, help="decode steps between two device polls")
    ap.add_argument("--tiny", action="store_true")
    args = ap.parse_args()
    cfg = EmmaXConfig.tiny() if args.tiny else EmmaXConfig.emma_x_7b()
    dev = "cuda:0"
    model = EmmaXForActionPrediction.from_synthetic(cfg, seed=5, device=dev, planted=True, max_batch=args.slots, max_prompt=64, max_ctx=1024)
    eng = model.engine
    rng = np.random.default_rng(3)
    N = args.requests
    ks = [int(k) for k in rng.integers(16, 480, size=N)]
    frames = torch.from_numpy(rng.integers(0, 256, size=(N, 224, 224, 3), dtype=np.uint8)).to(dev)
    rows = []
    for k in ks:
        r = [1] + [int(x) for x in rng.integers(3, 31744, size=31)]
        r[-1] = planted_start_token(cfg, k)
        rows.append(r)
    MAXN = 512
    # reference answers: every request alone (bs = 1); the planted chain is the a-priori answer while its margin holds
    want, n_chain = [], 0
    for i in range(N):
        ids, lens = model.generate_ids(rows[i:i + 1], frames_u8=frames[i:i + 1], max_new_tokens=MAXN)
        want.append(ids[0, : int(lens[0])].cpu().tolist())
        n_chain += int(want[-1] == planted_chain(cfg, rows[i][-1], 600))

    def sync():
        torch.cuda.synchronize()

    # ---- static batches
    sync(); t0 = time.perf_counter()
    got_static, lat_static = [], []
    for i in range(0, N, args.slots):
        ids, lens = model.generate_ids(rows[i:i + args.slots], frames_u8=frames[i:i + args.slots], max_new_tokens=MAXN)
        ids, lens = ids.cpu(), lens.cpu().tolist()
        t = time.perf_counter() - t0
        for b in range(len(lens)):
            got_static.append(ids[b, : lens[b]].tolist())
            lat_static.append(t)
    t_static = time.perf_counter() - t0

    def encode(fs):
        pe = eng.vision_encode(torch.stack(fs))
        return [pe[i] for i in range(len(fs))]

    def serve(trigger, after, overlap=True):
        sch = SlotScheduler(eng, encode, n_slots=args.slots, poll_every=args.poll, stop_trigger=trigger, stop_after=after,
                            encode_ahead=args.slots, overlap=overlap)
        sync(); t0 = time.perf_counter()
        for i in range(N):
            sch.submit(Request(i, frames[i], rows[i], MAXN))
        res = sch.run()
        sync()
        dt = time.perf_counter() - t0
        eng.set_stop([], 0)
        return dt, {r.rid: r for r in res}, sch

    serve([], 0)                                        # warm-up of both admission paths (allocator, first launches)
    t_block, res_block, sch_b = serve([], 0, overlap=False)   # round-3 behaviour: admissions on the decode stream
    t_cont, res_cont, sch_c = serve([], 0)
    t_early, res_early, sch_e = serve([29871], 8)
    # Agreement between the serving modes is REPORTED, not asserted, at this size: the launch plan picks tile geometry and
    # split-K factor from the problem size, so a frame encoded / prefilled alone and the same frame inside a batch of 8 see
    # different fp32 summation orders, and these synthetic weights keep their top-1 margin for ~80 steps only -- a near-tie
    # flips now and then (hundreds of steps x 32 requests).  The bit-exact comparison of a slot-served request with its own
    # bs = 1 run is the tiny-config GPU test (tests/test_serving_gpu.py), where the margins are wide.
    def cut_at_stop(ids):
        full = ids[:-1] if ids and ids[-1] == cfg.eos_token_id else ids
        return full[: next((t + 9 for t, v in enumerate(full) if v == 29871 and t + 9 <= len(full)), len(full))]

    assert sorted(res_cont) == list(range(N)) and sorted(res_early) == list(range(N))
    same_bs1_static = sum(int(got_static[i] == want[i]) for i in range(N))
    same_bs1_cont = sum(int(res_cont[i].ids == want[i]) for i in range(N))
    same_static_cont = sum(int(res_cont[i].ids == got_static[i]) for i in range(N))
    early_prefix = sum(int(res_early[i].ids == cut_at_stop(res_cont[i].ids)) for i in range(N))

    def pct(v, q):
        return float(np.percentile(np.asarray(v), q))

    lb = [res_block[i].latency_s for i in range(N)]
    lc = [res_cont[i].latency_s for i in range(N)]
    le = [res_early[i].latency_s for i in range(N)]
    out = {
        "model": "tiny" if args.tiny else "Emma-X-7B shapes (planted synthetic weights)", "requests": N, "slots": args.slots, "poll_every": args.poll,
        "new_tokens_to_eos": {"min": min(len(w) for w in want), "mean": float(np.mean([len(w) for w in want])), "max": max(len(w) for w in want)},
        "requests_following_the_planted_chain": n_chain,
        "static": {"seconds": round(t_static, 3), "actions_per_s": round(N / t_static, 3), "latency_p50_s": round(pct(lat_static, 50), 3),
                   "latency_p95_s": round(pct(lat_static, 95), 3)},
        "continuous_blocking_admission": {"seconds": round(t_block, 3), "actions_per_s": round(N / t_block, 3), "latency_p50_s": round(pct(lb, 50), 3),
                                          "latency_p95_s": round(pct(lb, 95), 3), "decode_steps": sch_b.steps},
        "continuous": {"admission": "overlapped (staging rows, second stream)", "overlapped_admissions": sch_c.overlapped_admissions, "seconds": round(t_cont, 3), "actions_per_s": round(N / t_cont, 3), "latency_p50_s": round(pct(lc, 50), 3),
                       "latency_p95_s": round(pct(lc, 95), 3), "decode_steps": sch_c.steps, "polls": sch_c.polls},
        "continuous_early_exit": {"seconds": round(t_early, 3), "actions_per_s": round(N / t_early, 3), "latency_p50_s": round(pct(le, 50), 3),
                                  "latency_p95_s": round(pct(le, 95), 3), "decode_steps": sch_e.steps},
        "requests_with_identical_ids": {"static_vs_bs1": same_bs1_static, "continuous_vs_bs1": same_bs1_cont,
                                        "continuous_vs_static": same_static_cont,
                                        "overlapped_vs_blocking_admission": sum(int(res_cont[i].ids == res_block[i].ids) for i in range(N)), "early_exit_is_prefix_of_continuous": early_prefix, "of": N},
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
