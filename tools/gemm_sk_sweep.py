"""Split-K sweep for the under-filled long-K projections of a short prefill (LLaMA-2-7B o-proj / down at 1-4 frames, M = 768 .. 3072):
128 x 128 against 256 x 256 tiles (tuning switch gemm_sk_big) over the slice counts, next to the launch plan (ksplit = 0), all inside one
process with the variants interleaved.  Prints us per launch (GEMM + reduce pass)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
import torch

from emmax import _lib as L

lib = L.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)   # 256 MB of scratch: every variant below fits
SHAPES = [("o", 4096, 4096), ("down", 4096, 11008)]
if os.environ.get("SK_SHAPES"):          # "name,N,K;..." e.g. the projector's fc2: "projfc2,4096,8704"
    SHAPES = [(t.split(",")[0], int(t.split(",")[1]), int(t.split(",")[2])) for t in os.environ["SK_SHAPES"].split(";")]
MS = [int(x) for x in os.environ.get("SK_MS", "768,1536,2304,3072").split(",")]
for name, N, K in SHAPES:
    for M in MS:
        A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        R = (torch.randn(M, N, device=dev) * 0.5).to(torch.bfloat16)
        variants = [("plan", -1, 0)] + [("small", 0, ks) for ks in (2, 3, 4, 8)] + [("big", 1, ks) for ks in (2, 3, 4, 5, 6, 8, 10, 16)]
        variants = [v for v in variants if v[2] == 0 or v[2] * M * N * 4 <= ws.numel() * 4]

        def run(v):
            L.tuning_set("gemm_sk_big", v[1])
            L.check(lib.emmax_op_gemm_splitk(A.data_ptr(), K, W.data_ptr(), K, C.data_ptr(), N, M, N, K, None, 0, None, R.data_ptr(), N, 0, v[2],
                                             ws.data_ptr(), ws.numel() * 4, st), "gemm_splitk")

        ref = None
        tot = {v: 0.0 for v in variants}
        for v in variants:
            run(v)
            torch.cuda.synchronize()
            if ref is None:
                ref = C.float().clone()
            else:
                assert (C.float() - ref).abs().max().item() <= 0.02 * ref.abs().max().item(), (name, M, v)
        nround, reps = 4, 10
        for rnd in range(nround):
            order = variants[rnd % len(variants):] + variants[:rnd % len(variants)]
            for v in order:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run(v)
                e1.record()
                torch.cuda.synchronize()
                tot[v] += e0.elapsed_time(e1) / reps / nround * 1e3
        L.tuning_set("gemm_sk_big", -1)
        big_tiles = ((M + 255) // 256) * (N // 256)
        print(f"{name} M={M} N={N} K={K} ({big_tiles} big tiles): " + "  ".join(f"{v[0]}{'/ks=%d' % v[2] if v[2] else ''} {tot[v]:.1f}" for v in variants), flush=True)
