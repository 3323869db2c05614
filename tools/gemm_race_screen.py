"""Race screen for the GEMM main-loop variants (lab, through the C ABI): every variant accumulates a tile's K steps in the same order, so the
staggered / deep-ring loops must reproduce the two-stage loop's output BIT FOR BIT -- on every repetition.  A DMA / fragment-read race
shows up as a rare differing tile.  usage: python tools/gemm_race_screen.py [reps [variants, e.g. -1,1,3]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
import torch

from emmax import _lib as L

SHAPES = [(256, 256, 64, 0), (256, 256, 128, 0), (512, 512, 192, 1), (4096, 4096, 4096, 0), (66816, 1024, 1024, 0), (8352, 4096, 1024, 1),
          (6144, 22016, 4096, 2), (768, 12288, 4096, 0), (65536, 1152, 4352, 0), (1000, 3456, 1152, 0), (8192, 8192, 8192, 0), (300, 384, 2176, 1)]


def main():
    global VARIANTS
    k32 = "--k32" in sys.argv   # round 5: the 128 x 256 x 32 tile (gemm_big = 2) against the 256 x 256 tile forced the same way (gemm_big = 1)
    if k32:
        sys.argv.remove("--k32")
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    VARIANTS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1, 1, 3]
    lib = L.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    bad = 0
    for M, N, K, act in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(M + N + K)
        A = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        bias = (torch.randn(N, device=dev, generator=g) * 0.1).to(torch.bfloat16) if act != 2 else None
        No = N // 2 if act == 2 else N

        def run(out):
            L.check(lib.emmax_op_gemm(A.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), No, M, N, K, L.ptr(bias), act, None, None, 0, 0, st), "gemm")

        L.tuning_set("gemm_deep", 0)
        if k32:
            L.tuning_set("gemm_deep", -1)
            L.tuning_set("gemm_big", 1)
        ref = torch.empty(M, No, dtype=torch.bfloat16, device=dev)
        run(ref)
        torch.cuda.synchronize()
        if k32:
            L.tuning_set("gemm_big", 2)
            nbad = 0
            for _ in range(reps):
                out = torch.full((M, No), float("nan"), dtype=torch.bfloat16, device=dev)
                run(out)
                torch.cuda.synchronize()
                nbad += int(not torch.equal(out.view(torch.int16), ref.view(torch.int16)))
            bad += nbad
            L.tuning_set("gemm_big", -1)
            print(f"M={M} N={N} K={K} act={act} k32 tile: {reps - nbad}/{reps} repetitions bit-identical to the 256 x 256 tile", flush=True)
            continue
        for deep in VARIANTS:
            L.tuning_set("gemm_deep", deep)
            nbad = 0
            for _ in range(reps):
                out = torch.full((M, No), float("nan"), dtype=torch.bfloat16, device=dev)
                run(out)
                torch.cuda.synchronize()
                if not torch.equal(out.view(torch.int16), ref.view(torch.int16)):
                    nbad += 1
            bad += nbad
            print(f"M={M} N={N} K={K} act={act} gemm_deep={deep}: {reps - nbad}/{reps} repetitions bit-identical to the two-stage loop", flush=True)
    L.tuning_set("gemm_deep", -1)
    print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad} differing outputs)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
