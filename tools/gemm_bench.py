"""GEMM micro-benchmark through the C ABI (emmax_op_gemm) on the shapes of the dense stages (not the headline bench):
ViT B=256 (DINOv2 261 tok, SigLIP 256 tok), projector, LLaMA prefill S=768 at B=1 and B=8.  Random normal operands."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
import torch

from emmax import _lib as L

SHAPES = [
    # name, M, N, K, act
    ("dino qkv", 261 * 256, 3072, 1024, 0), ("dino proj", 261 * 256, 1024, 1024, 0),
    ("dino fc1", 261 * 256, 4096, 1024, 1), ("dino fc2", 261 * 256, 1024, 4096, 0),
    ("siglip qkv", 256 * 256, 3456, 1152, 0), ("siglip proj", 256 * 256, 1152, 1152, 0),
    ("siglip fc1", 256 * 256, 4352, 1152, 1), ("siglip fc2", 256 * 256, 1152, 4352, 0),
    ("proj fc1", 256 * 256, 8704, 2176, 1), ("proj fc2", 256 * 256, 4096, 8704, 1),
    ("llama qkv B8", 768 * 8, 12288, 4096, 0), ("llama o B8", 768 * 8, 4096, 4096, 0),
    ("llama gateup B8", 768 * 8, 22016, 4096, 2), ("llama down B8", 768 * 8, 4096, 11008, 0),
    ("llama qkv B1", 768, 12288, 4096, 0), ("llama o B1", 768, 4096, 4096, 0),
    ("llama gateup B1", 768, 22016, 4096, 2), ("llama down B1", 768, 4096, 11008, 0),
    ("square 4096", 4096, 4096, 4096, 0), ("square 8192", 8192, 8192, 8192, 0),
]


def main():
    global SHAPES
    # --ab name=v0,v1[,v2]: every shape under each value of the tuning switch `name` (emmax_tuning_set), repetitions interleaved
    # in ONE process on ONE box (boxes differ by up to 20 %: numbers from different gpurun calls do not compare)
    ab = None
    argv = list(sys.argv[1:])
    if "--ab" in argv:
        i = argv.index("--ab")
        name, vals = argv[i + 1].split("=")
        ab = (name, [int(v) for v in vals.split(",")])
        del argv[i:i + 2]
    argv0 = list(argv)
    if "--interleave" in argv:
        argv.remove("--interleave")
    plan = "--session-plan" in argv   # launch plan of a session stage (64 MB split-K scratch: emmax_op_gemm_splitk with ksplit = 0)
    if plan:
        argv.remove("--session-plan")
    if argv:   # custom shapes: "M,N,K,act[,lda[,ldw[,ldc]]];..."  (row pitches of A, W, C in elements; 0 / absent: dense)
        SHAPES = [(f"custom{i}",) + tuple(int(v) for v in t.split(",")) for i, t in enumerate(argv[0].split(";"))]
    lib = L.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    if "--interleave" in argv0:
        # every listed shape allocated up front, then rounds over all of them in ROTATED order (clock / power state drifts inside a
        # process: consecutive one-shot timings of different variants do not compare); mean ms per shape over the rounds
        cases = []
        for name, M, N, K, act, *rest in SHAPES:
            rest = list(rest) + [0] * (4 - len(rest))
            lda, ldw, ldc = rest[0] or K, rest[1] or K, rest[2] or (N // 2 if act == 2 else N)
            epi = rest[3]      # 5th optional field: 1 = bias + LayerScale + in-place residual (the ViT proj / fc2 epilogue), 2 = bias only
            A = (torch.randn(M, lda, device=dev) * 0.5).to(torch.bfloat16)
            W = (torch.randn(N, ldw, device=dev) * 0.05).to(torch.bfloat16)
            C = (torch.randn(M, ldc, device=dev) * 0.5).to(torch.bfloat16)
            bias = torch.randn(N, device=dev).to(torch.bfloat16) if epi else None
            scale = torch.rand(N, device=dev).to(torch.bfloat16) if epi == 1 else None
            cases.append((name, M, N, K, act, lda, ldw, ldc, A, W, C, bias, scale, epi))

        def go(c):
            name, M, N, K, act, lda, ldw, ldc, A, W, C, bias, scale, epi = c
            L.check(lib.emmax_op_gemm(A.data_ptr(), lda, W.data_ptr(), ldw, C.data_ptr(), ldc, M, N, K, L.ptr(bias), act, L.ptr(scale),
                                      C.data_ptr() if epi == 1 else None, ldc, 0, st), "gemm")

        for c in cases:
            go(c)
        torch.cuda.synchronize()
        tot = [0.0] * len(cases)
        nround, reps = 3 * len(cases), 5
        for rnd in range(nround):
            order = list(range(len(cases)))
            order = order[rnd % len(cases):] + order[:rnd % len(cases)]
            for i in order:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    go(cases[i])
                e1.record()
                torch.cuda.synchronize()
                tot[i] += e0.elapsed_time(e1) / reps / nround
        for c, ms in zip(cases, tot):
            name, M, N, K, act, lda, ldw, ldc = c[:8]
            print(f"M={M:6d} N={N:6d} K={K:6d} act={act} epi={c[13]} ld={lda:5d},{ldw:5d},{ldc:5d}  {ms:8.4f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)
        return
    out = {}
    ws = torch.empty(16 << 20, dtype=torch.float32, device=dev) if plan else None
    for name, M, N, K, act, *rest in SHAPES:
        rest = list(rest) + [0] * (4 - len(rest))
        lda, ldw, ldc = rest[0] or K, rest[1] or K, rest[2] or (N // 2 if act == 2 else N)   # row pitches in elements (0: dense)
        A = (torch.randn(M, lda, device=dev) * 0.5).to(torch.bfloat16)
        W = (torch.randn(N, ldw, device=dev) * 0.05).to(torch.bfloat16)
        C = (torch.randn(M, ldc, device=dev) * 0.5).to(torch.bfloat16)
        epi = rest[3] if len(rest) > 3 else 0     # 1 = bias + LayerScale + in-place residual, 2 = bias
        bias = torch.randn(N, device=dev).to(torch.bfloat16) if epi else None
        scale = torch.rand(N, device=dev).to(torch.bfloat16) if epi == 1 else None

        def run():
            if plan:
                L.check(lib.emmax_op_gemm_splitk(A.data_ptr(), lda, W.data_ptr(), ldw, C.data_ptr(), ldc, M, N, K, None, act, None, None, 0, 0, 0,
                                                 ws.data_ptr(), ws.numel() * 4, st), "gemm")
                return
            L.check(lib.emmax_op_gemm(A.data_ptr(), lda, W.data_ptr(), ldw, C.data_ptr(), ldc, M, N, K, L.ptr(bias), act, L.ptr(scale),
                                      C.data_ptr() if epi == 1 else None, ldc, 0, st), "gemm")

        run()
        torch.cuda.synchronize()
        reps = 5
        if ab:
            tot = {v: 0.0 for v in ab[1]}
            for v in ab[1]:          # first launch of each variant (attribute set-up) outside the timing
                L.tuning_set(ab[0], v)
                run()
            # interleaved rounds of `reps` launches per variant, the order ROTATED from round to round: a variant that always ran behind a
            # slower (cooler) one measured 2-4 % faster than the identical code path in front of it (power management; round 4)
            nround = 2 * len(ab[1])
            for rnd in range(nround):
                for v in ab[1][rnd % len(ab[1]):] + ab[1][:rnd % len(ab[1])]:
                    L.tuning_set(ab[0], v)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    tot[v] += e0.elapsed_time(e1) / reps / nround
            tfs = {v: 2.0 * M * N * K / tot[v] / 1e9 for v in ab[1]}
            out[name] = {f"{ab[0]}={v}": round(tfs[v], 1) for v in ab[1]}
            print(f"{name:18s} M={M:6d} N={N:6d} K={K:6d} act={act}  " + "  ".join(f"{ab[0]}={v}: {tot[v]:8.4f} ms {tfs[v]:7.1f} TF/s" for v in ab[1]), flush=True)
            del A, W, C
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tf = 2.0 * M * N * K / ms / 1e9
        out[name] = {"ms": round(ms, 4), "tflops": round(tf, 1)}
        print(f"{name:18s} M={M:6d} N={N:6d} K={K:6d} act={act}  {ms:8.4f} ms  {tf:7.1f} TF/s" + (f"  ld={lda},{ldw},{ldc}" if any(rest) else ""), flush=True)
        del A, W, C
    print(json.dumps(out))


if __name__ == "__main__":
    main()
