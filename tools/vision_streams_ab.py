"""A/B of the vision encode with the two towers on one stream (vis_streams = 0) and side by side on two (the default up to 16 frames), alternating in
one process; checks that the patch embeddings are bit-identical.  Prints ms per call at 1 / 2 / 4 / 8 / 16 frames."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
import numpy as np
import torch

from emmax import _lib
from emmax.config import EmmaXConfig
from emmax.modeling import EmmaXForActionPrediction

cfg = EmmaXConfig.emma_x_7b()
model = EmmaXForActionPrediction.from_synthetic(cfg, seed=0, device="cuda:0", max_batch=int(os.environ.get("VIS_MAXB", "32")), max_prompt=512, max_ctx=1281)
eng = model.engine
rng = np.random.default_rng(0)
for B in [int(x) for x in os.environ.get("VIS_BS", "1,2,4,8,16,32").split(",")]:
    frames = torch.from_numpy(rng.integers(0, 256, size=(B, 224, 224, 3), dtype=np.uint8)).cuda()
    res, outs = {0: [], -1: []}, {}
    for rnd in range(3):
        for sw in (0, -1):
            with _lib.tuning(vis_streams=sw):
                outs[sw] = eng.vision_encode(frames).clone()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    eng.vision_encode(frames)
                e1.record()
                torch.cuda.synchronize()
                res[sw].append(e0.elapsed_time(e1) / 10)
    same = torch.equal(outs[0], outs[-1])
    print(f"B={B}: one stream {min(res[0]):.3f} ms ({', '.join('%.3f' % x for x in res[0])}) | two streams {min(res[-1]):.3f} ms ({', '.join('%.3f' % x for x in res[-1])}) | "
          f"patch embeddings {'IDENTICAL' if same else 'DIFFERENT'}", flush=True)
