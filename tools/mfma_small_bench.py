"""Tuning aid: time emmax_op_gemm_small / emmax_op_gemm_small_fp8 (the batch >= 3 / fp8 decode projection, MODE_PLAIN: no norm,
plain store, no stream-K workspace) over rotating weights, B = 1..8, for the four LLaMA-7B projection shapes.  The gap
between these and the fused decode stages of bench.py is the cost of the fused prologues / epilogues."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
from emmax import _lib as L
lib = L.load()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
FP8 = len(sys.argv) > 1 and sys.argv[1] == "fp8"
for N, K in [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)]:
    Ws = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(6)]
    if FP8:
        Wq = [torch.empty(N * K, dtype=torch.uint8, device=dev) for _ in Ws]
        Sc = [torch.empty(N, dtype=torch.float32, device=dev) for _ in Ws]
        for w, q, s in zip(Ws, Wq, Sc):
            L.check(lib.emmax_op_quant_fm8(w.data_ptr(), K, q.data_ptr(), s.data_ptr(), N, K, st), "quant")
    else:
        Wq = [torch.empty_like(w) for w in Ws]
        for w, f in zip(Ws, Wq):
            L.check(lib.emmax_op_repack_fm(w.data_ptr(), K, f.data_ptr(), N, K, st), "repack")
    del Ws
    line = []
    for B in (1, 2, 3, 4, 6, 8):
        x = torch.randn(B, K, device=dev).to(torch.bfloat16)
        y = torch.empty(B, N, dtype=torch.bfloat16, device=dev)
        def run():
            for i, f in enumerate(Wq):
                if FP8:
                    L.check(lib.emmax_op_gemm_small_fp8(x.data_ptr(), f.data_ptr(), Sc[i].data_ptr(), y.data_ptr(), B, N, K, st), "small8")
                else:
                    L.check(lib.emmax_op_gemm_small(x.data_ptr(), f.data_ptr(), y.data_ptr(), B, N, K, st), "small")
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        line.append("B=%d %.1f" % (B, e0.elapsed_time(e1) / 60 * 1e3))
    print(f"{'fp8' if FP8 else 'bf16'} N={N} K={K} ({N*K*(1 if FP8 else 2)/1e6:.0f} MB): " + "  ".join(line) + " us", flush=True)
