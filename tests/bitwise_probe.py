"""Run-to-run bit reproducibility probe: the full Emma-X-7B shape on random weights (seed 0), one packed
ragged prefill + 64 teacher-forced decode steps with FIXED token ids, at B = 1, 8, 32 and 64 (bf16-operand path: decode_ks / decode_km /
decode_kmp kernels, split-K and stream-K launches included), eager launches and hipGraph replay, plus the exact-numerics session at
B = 1, 2 and 8.  Every configuration runs TWICE in this process and must give bit-identical fp32 logits (sha256 over the prefill's
last-position rows and every decode step's rows); the hashes are printed as one JSON line so that tests/test_bitwise_gpu.py can compare
two PROCESSES.  Not a pytest module: started by that test (and by hand: `python tests/bitwise_probe.py`)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
import numpy as np
import torch

from emmax import _lib
from emmax.config import EmmaXConfig
from emmax.modeling import EmmaXForActionPrediction

T = 64


def run(model, B, graph, rng_seed):
    eng = model.engine
    rng = np.random.default_rng(rng_seed)
    frames = torch.from_numpy(rng.integers(0, 256, size=(B, 224, 224, 3), dtype=np.uint8)).cuda()
    lens = [512] + [int(x) for x in rng.integers(16, 513, size=B - 1)]
    rows = [[1] + [int(x) for x in rng.integers(3, 31744, size=n - 1)] for n in lens]
    forced = rng.integers(3, 31744, size=(T, B))
    h = hashlib.sha256()
    with _lib.tuning(graph=graph):
        model._prefill(rows, None, frames, max_new=T + 1)
        for t in range(T):
            h.update(eng.last_logits().float().cpu().numpy().tobytes())
            eng.set_current_tokens([int(x) for x in forced[t]])
            eng.decode_step()
        h.update(eng.last_logits().float().cpu().numpy().tobytes())
        assert eng.graph_active() == bool(graph)
    return h.hexdigest()


def main():
    cfg = EmmaXConfig.emma_x_7b()
    out = {}
    model = EmmaXForActionPrediction.from_synthetic(cfg, seed=0, device="cuda:0", max_batch=64, max_prompt=512, max_ctx=256 + 512 + T + 8)
    for B in (1, 8, 32, 64):
        for graph in (0, 1):
            a, b = run(model, B, graph, 100 + B), run(model, B, graph, 100 + B)
            assert a == b, f"B={B} graph={graph}: two runs in one process differ"
            out[f"bf16_B{B}_{'graph' if graph else 'eager'}"] = a
    del model
    torch.cuda.empty_cache()
    xm = EmmaXForActionPrediction.from_synthetic(EmmaXConfig.emma_x_7b(), seed=0, device="cuda:0", max_batch=8, max_prompt=512, max_ctx=256 + 512 + T + 8, exact=True)
    for B in (1, 2, 8):
        for graph in (0, 1):
            a, b = run(xm, B, graph, 200 + B), run(xm, B, graph, 200 + B)
            assert a == b, f"exact B={B} graph={graph}: two runs in one process differ"
            out[f"exact_B{B}_{'graph' if graph else 'eager'}"] = a
    print("BITWISE " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
