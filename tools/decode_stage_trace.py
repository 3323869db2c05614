"""Phase stamps of the batch-1 decode projections IN SITU (not part of the product): needs the library built with
-DDECODE_LAB_TRACE (tools/decode_stage_trace.sh builds it as tools/bin/libemmax_hip_trace.so and runs this with it copied over
the product library on the GPU box).  For every GEMV stage of the decode step the last launch of emmax_profile_decode_stage
(layer 31, weights HBM cold) is dissected: microseconds from the first block's entry to prologue done / activations staged
(barrier) / first weight block consumed / stream + epilogues done, min / median / max over the blocks."""
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
lib.emmax_debug_gemv_trace.restype = C.c_int
lib.emmax_debug_gemv_trace.argtypes = [C.c_void_p, C.c_int]
names = ["qkv_gemv", "paged_attn", "oproj_gemv", "gateup_gemv", "down_gemv", "lmhead_argmax"]
grids = {"qkv_gemv": 512, "oproj_gemv": 256, "gateup_gemv": 512, "down_gemv": 256, "lmhead_argmax": 512}
if B >= 3:   # the small-batch MFMA projections: when is each block's stream done, by XCD (block b runs on XCD b % 8)
    lib.emmax_debug_mfma_trace.restype = C.c_int
    lib.emmax_debug_mfma_trace.argtypes = [C.c_void_p, C.c_int]
    for rep in range(3):
        for i, n in enumerate(names):
            us = eng.profile_decode_stage(i, reps=2)
            if n == "paged_attn":
                continue
            buf = (C.c_ulonglong * (256 * 8))()
            assert lib.emmax_debug_mfma_trace(buf, 256 * 8) == 0
            tr = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.int64)
            t0 = tr[:, 0].min()
            done = (tr[:, 4] - t0) * 0.01
            print(f"{n}: {us:.2f} us per launch; stream done min / median / max {done.min():.1f} / {np.median(done):.1f} / {done.max():.1f}; by XCD mean:",
                  " ".join(f"{done[x::8].mean():5.1f}" for x in range(8)))
    sys.exit(0)
for i, n in enumerate(names):
    us = eng.profile_decode_stage(i, reps=2)
    if n == "paged_attn":
        print(f"{n}: {us:.2f} us per launch")
        continue
    buf = (C.c_ulonglong * (1024 * 8))()
    assert lib.emmax_debug_gemv_trace(buf, 1024 * 8) == 0
    tr = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
    t0s = tr[:, 0]
    nb = int((t0s > 0).sum()) if n != "lmhead_argmax" else 512
    nb = min(nb, grids[n])
    tr = tr[:nb]
    t0 = tr[:, 0].min()
    print(f"{n}: {us:.2f} us per launch, {nb} blocks; us from the first block's entry (min / median / max):")
    for k, lab in enumerate(["entry", "prologue done", "x staged (barrier)", "first block consumed", "stream + epilogues done"]):
        v = np.sort((tr[:, k] - t0) * 0.01)
        print(f"   {lab:26s} {v[0]:6.2f} / {v[nb // 2]:6.2f} / {v[-1]:6.2f}")
