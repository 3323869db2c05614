"""Stage timings on one MI355X (not the headline bench): vision encode at several batch sizes (BASELINE config 4 is
ViT-only B=256), prefill at S=768, with achieved TFLOP/s against the algorithmic FLOP counts of SURVEY.md 8d."""
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


def ev_time(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vision-batches", default="1,8,64,256")
    ap.add_argument("--prefill-batches", default="1,8")
    args = ap.parse_args()
    cfg = EmmaXConfig.emma_x_7b()
    vb = [int(x) for x in args.vision_batches.split(",")]
    model = EmmaXForActionPrediction.from_synthetic(cfg, seed=0, device="cuda:0", max_batch=max(vb), max_prompt=512, max_ctx=1281)
    eng = model.engine
    rng = np.random.default_rng(0)
    out = {"vision": {}, "prefill": {}}
    # algorithmic FLOPs per frame (towers run to block take_index; report against the full 420 GFLOP of SURVEY 8d too)
    def tower_flops(tw, blocks):
        N, D, M = tw.n_tokens, tw.embed_dim, tw.mlp_hidden
        return blocks * (2 * N * D * 3 * D + 4 * N * N * D + 2 * N * D * D + 4 * N * D * M) + 2 * 256 * 588 * D
    v, p1, h, _ = cfg.projector_dims
    proj = 2 * 256 * (v * p1 + p1 * h + h * h)
    executed = sum(tower_flops(t, t.take_index + 1) for t in cfg.towers) + proj
    nominal = sum(tower_flops(t, t.depth) for t in cfg.towers) + proj
    for B in vb:
        frames = torch.from_numpy(rng.integers(0, 256, size=(B, 224, 224, 3), dtype=np.uint8)).cuda()
        ms = ev_time(lambda: eng.vision_encode(frames), 5 if B <= 8 else 2)
        out["vision"][B] = {"ms": round(ms, 3), "frames_per_s": round(B / ms * 1e3, 1), "tflops_executed": round(executed * B / ms / 1e9, 1),
                            "tflops_vs_nominal_420g": round(nominal * B / ms / 1e9, 1)}
    L = cfg.llm
    for B in [int(x) for x in args.prefill_batches.split(",")]:
        frames = torch.from_numpy(rng.integers(0, 256, size=(B, 224, 224, 3), dtype=np.uint8)).cuda()
        prompts = [[1] + [int(x) for x in rng.integers(3, 31744, size=511)] for _ in range(B)]
        patches = eng.vision_encode(frames)
        eng.ensure_capacity(B, 512, 512)
        ms = ev_time(lambda: eng.prefill(prompts, patches), 3)
        S = 768
        flops = B * (2 * 6.6076e9 * S + 4 * S * S * 4096 * 32 / 2)
        out["prefill"][B] = {"ms": round(ms, 3), "tokens_per_s": round(B * S / ms * 1e3), "tflops": round(flops / ms / 1e9, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
