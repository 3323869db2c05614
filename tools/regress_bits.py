"""Bit-level regression probe between two builds of libemmax_hip.so (not part of the product): on the synthetic 7B model, patch
embeddings of 2 + 40 frames, the prefill logits of one 512-token prompt (all 768 rows), the first token and 24 greedy ids at batch 1, and
16 greedy ids of a ragged batch of 8 -- saved to /tmp/bits_<tag>.pt on the box, or compared with a saved file (`--against tag`)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
import numpy as np
import torch

from emmax.config import EmmaXConfig
from emmax.modeling import EmmaXForActionPrediction


def main():
    tag = sys.argv[1]
    against = sys.argv[sys.argv.index("--against") + 1] if "--against" in sys.argv else None
    cfg = EmmaXConfig.emma_x_7b()
    model = EmmaXForActionPrediction.from_synthetic(cfg, seed=0, device="cuda:0", max_batch=40, max_prompt=512, max_ctx=1281)
    eng = model.engine
    rng = np.random.default_rng(7)
    out = {}
    frames = torch.from_numpy(rng.integers(0, 256, size=(40, 224, 224, 3), dtype=np.uint8)).cuda()
    out["patches2"] = eng.vision_encode(frames[:2]).clone().cpu()
    out["patches40"] = eng.vision_encode(frames).clone().cpu()
    prompts = [[1] + [int(x) for x in rng.integers(3, 31744, size=n - 1)] for n in (512, 512, 480, 512, 300, 512, 64, 505)]
    patches = eng.vision_encode(frames[:8]).clone()
    eng.ensure_capacity(8, 512, 64)
    eng.prefill(prompts[:1], patches[:1])
    out["prefill_logits"] = eng.prefill_logits()[0].float().cpu()
    ids, lens = eng.generate(24, stop_on_eos=False)
    out["ids_b1"] = ids.cpu()
    eng.prefill(prompts, patches)
    ids, lens = eng.generate(16, stop_on_eos=False)
    out["ids_b8"] = ids.cpu()
    torch.save(out, f"/tmp/bits_{tag}.pt")     # (100 MB: box-local, not gpurun_out)
    if against:
        ref = torch.load(f"/tmp/bits_{against}.pt")
        for k in out:
            a, b = out[k], ref[k]
            same = torch.equal(a, b)
            extra = "" if same else f"  differing {(a != b).sum().item()} of {a.numel()}, max |d| {(a.float() - b.float()).abs().max().item():.3e} of max |ref| {b.float().abs().max().item():.3e}"
            print(f"{tag} vs {against}: {k:16s} {'IDENTICAL' if same else 'DIFFERENT'}{extra}", flush=True)


if __name__ == "__main__":
    main()
