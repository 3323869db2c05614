"""Phase stamps of the persistent layer chain IN SITU (not part of the product): needs the library built with -DDECODE_LAB_TRACE
(tools/decode_stage_trace.sh build; tools/chain_trace.sh runs this with it copied over the product library on the GPU box).
The last launch of emmax_profile_decode_stage(6) (layer 31: o-proj + gate/up + down + lm-head) is dissected per op: microseconds
from the first block's entry, min / median / max over the blocks, for wave 0 (does the epilogues) and wave 7."""
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
pe = eng.vision_encode(frames)
eng.prefill(ids, pe)
lib = _lib.load()
lib.emmax_debug_ks_trace.restype = C.c_int
lib.emmax_debug_ks_trace.argtypes = [C.c_void_p, C.c_int]
for rep in range(2):
    us = eng.profile_decode_stage(6, reps=2)
    n = 512 * 2 * 4 * 6
    buf = (C.c_ulonglong * n)()
    assert lib.emmax_debug_ks_trace(buf, n) == 0
    tr = np.frombuffer(buf, dtype=np.uint64).reshape(512, 2, 4, 6).astype(np.int64)
    t0 = tr[:, :, 0, 0].min()
    print(f"chain: {us:.2f} us per launch (rep {rep}); us from the first block's entry, min / median / max over 512 blocks")
    for op, name in enumerate(["o-proj", "gate/up", "down", "tail (lm-head)"]):
        for w, wn in enumerate(["wave0", "wave7"]):
            row = []
            for k, lab in enumerate(["enter", "x ready", "stale polls", "stream done", "barrier", "epilogue"]):
                v = tr[:, w, op, k]
                v = np.sort(v if k == 2 else (v - t0) * 0.01)
                row.append(f"{lab} {v[0]:.1f}/{v[256]:.1f}/{v[-1]:.1f}")
            print(f"  {name:14s} {wn}: " + " | ".join(row))
