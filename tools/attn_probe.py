"""Workload for rocprofv3 passes over the prefill/ViT attention kernel alone: the three shapes of
profiles/r01_vision_prefill_kernel_stats.csv through emmax_op_attention (DINOv2 B=256 N=261 hd 64; SigLIP B=256 N=256 hd 72;
LLaMA prefill causal B=8 S=768 hd 128).  Prints HIP-event timings and TFLOP/s (4*N^2*D*H per sequence; causal counted in full
like the r01 figures, plus the executed half in parentheses)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
import torch

from emmax import _lib

lib = _lib.load()
dev = "cuda:0"
REPS = int(os.environ.get("ATTN_REPS", "20"))
shapes = [("llama<128> causal B=1", 1, 768, 32, 128, 1), ("dino<64>", 256, 261, 16, 64, 0), ("siglip<72>", 256, 256, 16, 72, 0), ("llama<128> causal", 8, 768, 32, 128, 1)]
for name, B, N, H, hd, causal in shapes:
    D = H * hd
    ld = (3 * D + 127) // 128 * 128
    qkv = (torch.randn(B * N, ld, device=dev) * 0.5).to(torch.bfloat16)
    out = torch.empty(B * N, D, dtype=torch.bfloat16, device=dev)
    cu = torch.arange(0, (B + 1) * N, N, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.check(lib.emmax_op_attention(qkv.data_ptr(), ld, 0, D, 2 * D, out.data_ptr(), D, cu.data_ptr(), B, N, H, H, hd, hd ** -0.5, causal, st))

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / REPS * 1e3
    fl = 4.0 * N * N * hd * H * B
    print(f"{name}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s" + (f" ({fl / 2 / us / 1e6:.0f} executed)" if causal else ""))
