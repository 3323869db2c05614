"""Minimal workload for rocprofv3 --pmc passes (counter collection serialises every dispatch, so the full bench is far
too slow under it): full-size LLaMA layer dims but only 2 layers / 2-block towers, one prefill, then a few launches of
each decode stage.  The decode kernels and their launch shapes are exactly those of the 7B run (B=1)."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
import numpy as np
import torch

from emmax.config import EmmaXConfig
from emmax.modeling import EmmaXForActionPrediction

cfg = EmmaXConfig.emma_x_7b()
cfg.llm.num_layers = 2
for t in cfg.towers:
    t.depth = 3
B = int(os.environ.get("PROBE_BATCH", "1"))
if os.environ.get("PROBE_FP8"):   # e4m3 decode weights (BASELINE configs[4])
    cfg.decode_weight_dtype = "fp8"
EXACT = int(os.environ.get("PROBE_EXACT", "0"))   # 1: exact numerics (24-bit K / V cache), 2: with fp32 K / V rows
model = EmmaXForActionPrediction.from_synthetic(cfg, seed=0, device="cuda:0", max_batch=B, max_prompt=512, max_ctx=1281, exact=EXACT)
rng = np.random.default_rng(0)
frames = torch.from_numpy(rng.integers(0, 256, size=(B, 224, 224, 3), dtype=np.uint8)).cuda()
prompts = [[1] + [int(x) for x in rng.integers(3, 31744, size=511)] for _ in range(B)]
model._prefill(prompts, None, frames, max_new=8)
for stage in range(6):
    us = model.engine.profile_decode_stage(stage, reps=2)
    print("stage", stage, "us", us)
torch.cuda.synchronize()
