"""Tuning aid: time emmax_op_gemv (the batch <= 2 decode projection, MODE_PLAIN: no norm, plain store) on rotating weights."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
from emmax import _lib as L
lib = L.load()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
for B, N, K in [(1, 4096, 11008), (1, 12288, 4096), (1, 22016, 4096), (1, 4096, 4096), (2, 4096, 11008)]:
    x = torch.randn(B, K, device=dev).to(torch.bfloat16)
    Ws = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(8)]
    y = torch.empty(B, N, dtype=torch.bfloat16, device=dev)
    def run():
        for w in Ws:
            L.check(lib.emmax_op_gemv(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, N, K, st), "gemv")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 160 * 1e3
    print(f"B={B} N={N} K={K}: {us:.1f} us  {N*K*2/us/1e6:.2f} TB/s", flush=True)
