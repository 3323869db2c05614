"""Tuning aid: time emmax_op_gemm_small (the batch >= 3 decode projection, MODE_PLAIN) for a few shapes."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
from emmax import _lib as L
lib = L.load()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
for B, N, K in [(8, 4096, 11008), (8, 4096, 4096), (8, 12288, 4096), (8, 32064, 4096), (3, 4096, 11008)]:
    x = (torch.randn(B, K, device=dev)).to(torch.bfloat16)
    Ws = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(6)]
    Wfm = [torch.empty_like(w) for w in Ws]
    for w, f in zip(Ws, Wfm):
        L.check(lib.emmax_op_repack_fm(w.data_ptr(), K, f.data_ptr(), N, K, st), "repack")
    y = torch.empty(B, N, dtype=torch.bfloat16, device=dev)
    def run():
        for f in Wfm:
            L.check(lib.emmax_op_gemm_small(x.data_ptr(), f.data_ptr(), y.data_ptr(), B, N, K, st), "small")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 60 * 1e3
    print(f"B={B} N={N} K={K}: {us:.1f} us  {N*K*2/us/1e6:.2f} TB/s", flush=True)
