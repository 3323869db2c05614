"""A/B of the causal prefill attention's two block forms (tuning switch attn_ksplit): four query waves
per block (0) against eight waves = two key groups over the same four query waves (1), alternating inside one process, at the
LLaMA-2-7B prefill shape (32 heads, head_dim 128) for 1 / 2 / 4 / 8 sequences of 768 tokens and one of 1024."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
import torch

from emmax import _lib

lib = _lib.load()
dev = "cuda:0"
REPS = int(os.environ.get("ATTN_REPS", "50"))
H, hd = 32, 128
for B, N in [(1, 768), (1, 1024), (1, 300), (2, 768), (4, 768), (8, 768)]:
    D = H * hd
    ld = 3 * D
    qkv = (torch.randn(B * N, ld, device=dev) * 0.5).to(torch.bfloat16)
    out = torch.empty(B * N, D, dtype=torch.bfloat16, device=dev)
    cu = torch.arange(0, (B + 1) * N, N, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.check(lib.emmax_op_attention(qkv.data_ptr(), ld, 0, D, 2 * D, out.data_ptr(), D, cu.data_ptr(), B, N, H, H, hd, hd ** -0.5, 1, st))

    res = {0: [], 1: []}
    outs = {}
    for rnd in range(3):
        for sw in (0, 1):
            with _lib.tuning(attn_ksplit=sw):
                run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(REPS):
                    run()
                e1.record()
                torch.cuda.synchronize()
                res[sw].append(e0.elapsed_time(e1) / REPS * 1e3)
                outs[sw] = out.float().clone()
    d = (outs[0] - outs[1]).abs().max().item()
    print(f"B={B} S={N} blocks={B * H * ((N + 127) // 128)}: four query waves {min(res[0]):.1f} us ({', '.join('%.1f' % x for x in res[0])}) | "
          f"two key groups {min(res[1]):.1f} us ({', '.join('%.1f' % x for x in res[1])}) | max |difference| of the outputs {d:.3e}")
