"""Phase stamps of the stand-alone K-split projections IN SITU (lab; needs the -DDECODE_LAB_TRACE library, see
tools/lab_trace.sh): for every GEMV stage the last launch of emmax_profile_decode_stage (layer 31) is dissected -- us from the
first block's entry: min / median / max over the blocks, waves 0 (epilogue wave) and 7."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
import numpy as np
import torch

from emmax.config import EmmaXConfig
from emmax.modeling import EmmaXForActionPrediction
from emmax import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = EmmaXConfig.emma_x_7b()
model = EmmaXForActionPrediction.from_synthetic(cfg, seed=0, device="cuda:0", max_batch=B, max_prompt=512, max_ctx=1281)
eng = model.engine
rng = np.random.default_rng(0)
frames = torch.from_numpy(rng.integers(0, 256, size=(B, 224, 224, 3), dtype=np.uint8)).to("cuda:0")
ids = [list(rng.integers(3, 32000, size=512)) for _ in range(B)]
eng.prefill(ids, eng.vision_encode(frames))
lib = _lib.load()
lib.emmax_debug_ks_trace.restype = C.c_int
lib.emmax_debug_ks_trace.argtypes = [C.c_void_p, C.c_int]
names = {0: "qkv", 2: "o-proj", 3: "gate/up", 4: "down", 5: "lm-head"}
grids = {0: 512, 2: 256, 3: 512, 4: 512, 5: 512}
for st, name in names.items():
    us = eng.profile_decode_stage(st, reps=2)
    n = 512 * 2 * 4 * 6
    buf = (C.c_ulonglong * n)()
    assert lib.emmax_debug_ks_trace(buf, n) == 0
    tr = np.frombuffer(buf, dtype=np.uint64).reshape(512, 2, 4, 6).astype(np.int64)[: grids[st]]
    t0 = tr[:, :, 0, 0].min()
    nb = grids[st]
    print(f"{name}: {us:.2f} us per launch; us from the first block's entry, min / median / max over {nb} blocks; second-half blocks (dispatched onto an occupied CU) separately")
    for w, wn in enumerate(["wave0", "wave7"]):
        row = []
        for k, lab in enumerate(["enter", "x ready", None, "stream done", "barrier", "epilogue"]):
            if lab is None:
                continue
            v = np.sort((tr[:, w, 0, k] - t0) * 0.01)
            row.append(f"{lab} {v[0]:.1f}/{v[nb // 2]:.1f}/{v[-1]:.1f}")
        print(f"   {wn}: " + " | ".join(row))
    if nb == 512:
        for half, hn in ((slice(0, 256), "blocks 0-255  "), (slice(256, 512), "blocks 256-511")):
            v = (tr[half, 1, 0, :] - t0) * 0.01
            print(f"   {hn} wave7 median: enter {np.median(v[:, 0]):.1f} x ready {np.median(v[:, 1]):.1f} stream done {np.median(v[:, 3]):.1f} barrier {np.median(v[:, 4]):.1f}")
