"""Per-kernel parity on a real MI355X: each HIP kernel (called through the C ABI) vs the CPU oracle's op in fp32 on the
same bf16-rounded inputs.  Tolerances are bf16-output tolerances: |err| <= TOL * max|ref| with TOL = 1e-2 unless noted
(one bf16 rounding is 2^-9 = 2e-3 relative; fp32 accumulation order adds ~1e-6)."""

import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-2


def _lib():
    from emmax import _lib

    return _lib, _lib.load()


def bf(x):
    return x.to(torch.bfloat16)


def relerr(got, ref):
    ref = ref.float().cpu()
    return ((got.float().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()


def assert_elementwise(got, ref, rtol=1e-2, atol_frac=4e-3, what=""):
    """Per-ELEMENT bound |err| <= atol + rtol * |ref| with atol = atol_frac * rms(ref): the global `relerr` (max|err| / max|ref|)
    cannot see a kernel that is wrong only on small-magnitude outputs -- an error of 1 % of max|ref| on an element whose true
    value is 0.1 % of it passes there and fails here.  rtol 1e-2 = a bf16 output rounding (2^-9) plus a rounded epilogue operand;
    atol = what fp32 accumulation order and one bf16 rounding of a typical-magnitude term leave on an output that cancels to ~0
    (atol_frac 4e-3: two bf16 ulps of the rms)."""
    g, r = got.float().cpu(), ref.float().cpu()
    assert torch.isfinite(g).all(), what
    bound = atol_frac * r.pow(2).mean().sqrt().item() + rtol * r.abs()
    bad = (g - r).abs() > bound
    if bad.any():
        i = int(((g - r).abs() - bound).argmax())
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.numel()} elements outside atol + rtol|ref|; worst at flat index {i}: "
                             f"got {g.flatten()[i].item():.6g} ref {r.flatten()[i].item():.6g} bound {bound.flatten()[i].item():.3g}")


# prefill / ViT attention: P is rounded to bf16 for the PV MFMA (as in the reference's bf16 SDPA), so an output that cancels to ~0
# carries 2^-9 x the magnitude of its terms: measured up to 1.0e-2 x rms(ref) -- still 5x tighter than 1e-2 x max|ref|
ATTN_ATOL = 2e-2


def stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("M,N,K", [(261, 128, 64), (128, 256, 128), (300, 384, 640), (1, 128, 64), (517, 1152, 1024)])
@pytest.mark.parametrize("variant", ["plain", "bias", "gelu", "scale_res", "f32"])
def test_gemm(device, M, N, K, variant):
    L, lib = _lib()
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    bias = bf(torch.randn(N, generator=g)) if variant != "plain" else None
    scale = bf(torch.rand(N, generator=g) + 0.5) if variant == "scale_res" else None
    res = bf(torch.randn(M, N, generator=g)) if variant == "scale_res" else None
    act = 1 if variant == "gelu" else 0
    out_f32 = variant == "f32"
    ref = A.float() @ W.float().t()
    if bias is not None:
        ref = ref + bias.float()
    if act:
        ref = F.gelu(ref)
    if scale is not None:
        ref = ref * scale.float() + res.float()
    Ad, Wd = A.to(device), W.to(device)
    Cd = torch.full((M, N), float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16, device=device)
    bd = bias.to(device) if bias is not None else None
    sd = scale.to(device) if scale is not None else None
    rd = res.to(device) if res is not None else None
    L.check(lib.emmax_op_gemm(Ad.data_ptr(), K, Wd.data_ptr(), K, Cd.data_ptr(), N, M, N, K, L.ptr(bd), act, L.ptr(sd), L.ptr(rd), N,
                              int(out_f32), stream()), "gemm")
    torch.cuda.synchronize()
    assert torch.isfinite(Cd.float()).all()
    assert relerr(Cd, ref) < (1e-4 if out_f32 else TOL)
    assert_elementwise(Cd, ref, rtol=1e-4 if out_f32 else 1e-2, atol_frac=1e-4 if out_f32 else 4e-3)


def test_gemm_swiglu(device):
    L, lib = _lib()
    M, inter, K = 200, 192, 128
    g = torch.Generator().manual_seed(3)
    A = bf(torch.randn(M, K, generator=g))
    Wg = bf(torch.randn(inter, K, generator=g) * 0.1)
    Wu = bf(torch.randn(inter, K, generator=g) * 0.1)
    ref = F.silu(A.float() @ Wg.float().t()) * (A.float() @ Wu.float().t())
    # 16-row interleave: [g0..15, u0..15, g16..31, u16..31, ...]
    Wi = torch.stack([Wg.view(inter // 16, 16, K), Wu.view(inter // 16, 16, K)], dim=1).reshape(2 * inter, K).contiguous()
    Ad, Wd = A.to(device), Wi.to(device)
    Cd = torch.zeros(M, inter, dtype=torch.bfloat16, device=device)
    L.check(lib.emmax_op_gemm(Ad.data_ptr(), K, Wd.data_ptr(), K, Cd.data_ptr(), inter, M, 2 * inter, K, None, 2, None, None, 0, 0,
                              stream()), "gemm swiglu")
    torch.cuda.synchronize()
    assert relerr(Cd, ref) < TOL
    assert_elementwise(Cd, ref)


@pytest.mark.parametrize("geom", [-1, 2], ids=["planned", "k32"])
@pytest.mark.parametrize("variant", ["plain", "gelu", "scale_res", "f32", "swiglu"])
def test_gemm_big_tile(device, tune, variant, geom):
    """>= 512 tiles of 256x256 route to the 256x256x64 geometry of the GEMM kernel (gemm.hip): ragged M (4100 = 16 tiles + 4
    rows), N = 65 * 128 (the last tile column is half empty), every epilogue; reference = fp32 matmul on the same device."""
    L, lib = _lib()
    tune(gemm_big=geom)   # 2: the 128 x 256 x 32 tile of round 5 (K = 192 = six 32-deep steps: the three-stage ring wraps twice)
    M, N, K = 4100, 8320, 192
    g = torch.Generator().manual_seed(11)
    A = bf(torch.randn(M, K, generator=g)).to(device)
    W = bf(torch.randn(N, K, generator=g) * 0.05).to(device)
    ref = A.float() @ W.float().t()
    if variant == "swiglu":
        inter = N // 2
        Wg, Wu = W[:inter], W[inter:]
        ref = F.silu(A.float() @ Wg.float().t()) * (A.float() @ Wu.float().t())
        Wi = torch.stack([Wg.view(inter // 16, 16, K), Wu.view(inter // 16, 16, K)], dim=1).reshape(N, K).contiguous()
        Cd = torch.full((M, inter), float("nan"), dtype=torch.bfloat16, device=device)
        L.check(lib.emmax_op_gemm(A.data_ptr(), K, Wi.data_ptr(), K, Cd.data_ptr(), inter, M, N, K, None, 2, None, None, 0, 0, stream()), "gemm")
        torch.cuda.synchronize()
        assert torch.isfinite(Cd.float()).all()
        assert relerr(Cd, ref) < TOL
        assert_elementwise(Cd, ref)
        return
    bias = bf(torch.randn(N, generator=g)).to(device) if variant != "plain" else None
    scale = bf(torch.rand(N, generator=g) + 0.5).to(device) if variant == "scale_res" else None
    res = bf(torch.randn(M, N, generator=g)).to(device) if variant == "scale_res" else None
    if bias is not None:
        ref = ref + bias.float()
    if variant == "gelu":
        ref = F.gelu(ref)
    if scale is not None:
        ref = ref * scale.float() + res.float()
    out_f32 = variant == "f32"
    Cd = torch.full((M, N), float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16, device=device)
    L.check(lib.emmax_op_gemm(A.data_ptr(), K, W.data_ptr(), K, Cd.data_ptr(), N, M, N, K, L.ptr(bias), int(variant == "gelu"),
                              L.ptr(scale), L.ptr(res), N, int(out_f32), stream()), "gemm")
    torch.cuda.synchronize()
    assert torch.isfinite(Cd.float()).all()
    assert relerr(Cd, ref) < (1e-4 if out_f32 else TOL)
    assert_elementwise(Cd, ref, rtol=1e-4 if out_f32 else 1e-2, atol_frac=1e-4 if out_f32 else 4e-3)


@pytest.mark.parametrize("variant", ["plain", "scale_res", "f32"])
def test_gemm_column_split_plan(device, variant):
    """M = 128 big tile rows, N = 1152 = 4.5 big tile columns: the launch plan sends columns 0..1023 to the 256x256 geometry
    and the half-empty last tile column to a separate 128x128 launch (bias / LayerScale / residual / C offsets per part)."""
    L, lib = _lib()
    M, N, K = 32768, 1152, 128
    g = torch.Generator().manual_seed(29)
    A = bf(torch.randn(M, K, generator=g)).to(device)
    W = bf(torch.randn(N, K, generator=g) * 0.05).to(device)
    ref = A.float() @ W.float().t()
    bias = bf(torch.randn(N, generator=g)).to(device) if variant != "plain" else None
    scale = bf(torch.rand(N, generator=g) + 0.5).to(device) if variant == "scale_res" else None
    res = bf(torch.randn(M, N, generator=g)).to(device) if variant == "scale_res" else None
    if bias is not None:
        ref = ref + bias.float()
    if scale is not None:
        ref = ref * scale.float() + res.float()
    out_f32 = variant == "f32"
    Cd = torch.full((M, N), float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16, device=device)
    L.check(lib.emmax_op_gemm(A.data_ptr(), K, W.data_ptr(), K, Cd.data_ptr(), N, M, N, K, L.ptr(bias), 0, L.ptr(scale), L.ptr(res), N,
                              int(out_f32), stream()), "gemm")
    torch.cuda.synchronize()
    assert torch.isfinite(Cd.float()).all()
    assert relerr(Cd, ref) < (1e-4 if out_f32 else TOL)
    assert_elementwise(Cd, ref, rtol=1e-4 if out_f32 else 1e-2, atol_frac=1e-4 if out_f32 else 4e-3)


@pytest.mark.parametrize("M,N,K,ks", [(768, 4096, 4096, 2), (768, 4096, 11008, 4), (261, 1024, 4352, 8), (300, 384, 640, 3)])
@pytest.mark.parametrize("variant", ["plain", "gelu", "scale_res_inplace", "f32"])
def test_gemm_splitk(device, M, N, K, ks, variant):
    """K slices per tile + reduce / epilogue pass (under-filled problems with a long K); incl. a slice count that does not
    divide the K steps and the in-place residual form the prefill uses (C == residual)."""
    _gemm_splitk_case(device, M, N, K, ks, variant)


@pytest.mark.parametrize("M,N,K,ks", [(768, 4096, 11008, 5), (1536, 4096, 11008, 2), (300, 384, 640, 3), (1000, 4096, 8704, 4)])
@pytest.mark.parametrize("variant", ["plain", "scale_res_inplace", "f32"])
def test_gemm_splitk_on_big_tiles(device, M, N, K, ks, variant):
    """The same pass with the K slices taken from 256 x 256 tiles (tuning switch gemm_sk_big; what the launch plan picks for the down
    projection of a one- / two-frame prefill: 5 / 2 slices), incl. ragged edge tiles and a slice count that does not divide the K steps."""
    L, _ = _lib()
    with L.tuning(gemm_sk_big=1):
        _gemm_splitk_case(device, M, N, K, ks, variant)


def _gemm_splitk_case(device, M, N, K, ks, variant):
    L, lib = _lib()
    g = torch.Generator().manual_seed(M + N + K + ks)
    A = bf(torch.randn(M, K, generator=g)).to(device)
    W = bf(torch.randn(N, K, generator=g) * 0.03).to(device)
    ref = A.float() @ W.float().t()
    bias = bf(torch.randn(N, generator=g)).to(device) if variant != "plain" else None
    scale = bf(torch.rand(N, generator=g) + 0.5).to(device) if variant == "scale_res_inplace" else None
    out_f32 = variant == "f32"
    if bias is not None:
        ref = ref + bias.float()
    if variant == "gelu":
        ref = F.gelu(ref)
    if scale is not None:
        res0 = bf(torch.randn(M, N, generator=g)).to(device)
        ref = ref * scale.float() + res0.float()
        Cd = res0.clone()                     # C doubles as the residual
        res = Cd
    else:
        Cd = torch.full((M, N), float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16, device=device)
        res = None
    ws = torch.empty(ks * M * N, dtype=torch.float32, device=device)
    L.check(lib.emmax_op_gemm_splitk(A.data_ptr(), K, W.data_ptr(), K, Cd.data_ptr(), N, M, N, K, L.ptr(bias), int(variant == "gelu"),
                                     L.ptr(scale), L.ptr(res), N, int(out_f32), ks, ws.data_ptr(), ws.numel() * 4, stream()), "splitk")
    torch.cuda.synchronize()
    assert torch.isfinite(Cd.float()).all()
    assert relerr(Cd, ref) < (1e-4 if out_f32 else TOL)
    assert_elementwise(Cd, ref, rtol=1e-4 if out_f32 else 1e-2, atol_frac=1e-4 if out_f32 else 4e-3)
    # too small a workspace / too many slices are errors, not silent fallbacks
    assert lib.emmax_op_gemm_splitk(A.data_ptr(), K, W.data_ptr(), K, Cd.data_ptr(), N, M, N, K, None, 0, None, None, N, 0, ks,
                                    ws.data_ptr(), 1024, stream()) != 0
    assert lib.emmax_op_gemm_splitk(A.data_ptr(), K, W.data_ptr(), K, Cd.data_ptr(), N, M, N, K, None, 0, None, None, N, 0, K // 64 + 1,
                                    ws.data_ptr(), ws.numel() * 4, stream()) != 0


def _swiglu_case(device, M, inter, K, seed):
    g = torch.Generator().manual_seed(seed)
    A = bf(torch.randn(M, K, generator=g)).to(device)
    Wg = bf(torch.randn(inter, K, generator=g) * 0.03).to(device)
    Wu = bf(torch.randn(inter, K, generator=g) * 0.03).to(device)
    ref = F.silu(A.float() @ Wg.float().t()) * (A.float() @ Wu.float().t())
    Wi = torch.stack([Wg.view(inter // 16, 16, K), Wu.view(inter // 16, 16, K)], dim=1).reshape(2 * inter, K).contiguous()
    return A, Wi, ref


@pytest.mark.parametrize("M,inter,K,ks", [(768, 128, 4096, 8), (300, 192, 640, 3), (768, 2048, 2048, 2)])
def test_gemm_splitk_swiglu(device, M, inter, K, ks):
    """SwiGLU through the split-K path: fp32 partial tiles in the interleaved (gate, up) column order, pairing in the reduce pass."""
    L, lib = _lib()
    A, Wi, ref = _swiglu_case(device, M, inter, K, M + inter + K)
    N = 2 * inter
    Cd = torch.full((M, inter), float("nan"), dtype=torch.bfloat16, device=device)
    ws = torch.empty(ks * M * N, dtype=torch.float32, device=device)
    L.check(lib.emmax_op_gemm_splitk(A.data_ptr(), K, Wi.data_ptr(), K, Cd.data_ptr(), inter, M, N, K, None, 2, None, None, 0, 0, ks,
                                     ws.data_ptr(), ws.numel() * 4, stream()), "splitk swiglu")
    torch.cuda.synchronize()
    assert torch.isfinite(Cd.float()).all()
    assert relerr(Cd, ref) < TOL
    assert_elementwise(Cd, ref)
    # the same launch without the split: equal up to the fp32 summation order
    C1 = torch.empty_like(Cd)
    L.check(lib.emmax_op_gemm(A.data_ptr(), K, Wi.data_ptr(), K, C1.data_ptr(), inter, M, N, K, None, 2, None, None, 0, 0, stream()), "gemm")
    torch.cuda.synchronize()
    assert relerr(Cd, C1.float()) < 4e-3


@pytest.mark.parametrize("act", [2, 0])
def test_gemm_hybrid_column_remainder_plan(device, act):
    """One-frame prefill gate/up (M = 768, N = 22016 = 86 tile columns, K = 4096): 3 x 86 = 258 big tiles = one round + 2 tiles.
    The session's launch plan (emmax_op_gemm_splitk with ksplit = 0) runs 85 tile columns as one round of 256x256 tiles and the last
    256 columns K-split on 128x128 tiles + the reduce / epilogue pass; against fp32, and against the plan with the switch off."""
    L, lib = _lib()
    M, N, K = 768, 22016, 4096
    if act == 2:
        A, W, ref = _swiglu_case(device, M, N // 2, K, 5)
        nout = N // 2
        res = None
    else:
        g = torch.Generator().manual_seed(6)
        A = bf(torch.randn(M, K, generator=g)).to(device)
        W = bf(torch.randn(N, K, generator=g) * 0.03).to(device)
        res = bf(torch.randn(M, N, generator=g)).to(device)
        ref = A.float() @ W.float().t() + res.float()
        nout = N
    ws = torch.empty(16 << 20, dtype=torch.float32, device=device)
    outs = []
    for hybrid in (1, 0):
        with L.tuning(gemm_hybrid=hybrid):
            Cd = torch.full((M, nout), float("nan"), dtype=torch.bfloat16, device=device)
            L.check(lib.emmax_op_gemm_splitk(A.data_ptr(), K, W.data_ptr(), K, Cd.data_ptr(), nout, M, N, K, None, act, None, L.ptr(res), N, 0, 0,
                                             ws.data_ptr(), ws.numel() * 4, stream()), "planned gemm")
            torch.cuda.synchronize()
        assert torch.isfinite(Cd.float()).all()
        assert relerr(Cd, ref) < TOL
        assert_elementwise(Cd, ref)
        outs.append(Cd)
    # whole-tile columns: the K order of an output element does not depend on the tile geometry; the K-split columns differ from the
    # unsplit launch by the fp32 summation order only
    n1 = 85 * 256 // (2 if act == 2 else 1)
    assert torch.equal(outs[0][:, :n1], outs[1][:, :n1])
    assert relerr(outs[0][:, n1:], outs[1][:, n1:].float()) < 4e-3


@pytest.mark.parametrize("geom", [-1, 2], ids=["planned", "k32"])
@pytest.mark.parametrize("M,N,K,act", [(261 * 3, 3072, 1024, 0), (256 * 5 + 17, 4352, 1152, 1), (40, 384, 128, 0), (66816 // 8, 4096, 1024, 1)])
def test_gemm_with_layernorm_folded_in(device, tune, M, N, K, act, geom):
    """timm Block: norm1 -> attn.qkv and norm2 -> mlp.fc1 as ONE GEMM over the raw rows (W' = bf16(W .* gamma), row statistics from a
    stats-only pass, y = rstd (acc - mean * colsum(W')) + (W beta + bias) in the epilogue) against LayerNorm + linear (+ exact-erf
    GELU) in fp32.  Rows carry a large common offset and a few massive channels (what ViT token streams look like): the mean term
    must cancel exactly, not approximately.  Every launch plan: big tiles, small tiles, row split, column split (N = 4352)."""
    L_, lib = _lib()
    tune(gemm_big=geom)
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g) * 0.7 + 3.0 * torch.randn(M, 1, generator=g)
    x[:, 5] += 40.0
    x[::7, 100] -= 25.0
    x = bf(x)
    W = bf(torch.randn(N, K, generator=g) * 0.03)
    gamma = bf(1.0 + 0.3 * torch.randn(K, generator=g))
    beta = bf(0.1 * torch.randn(K, generator=g))
    bias = bf(0.05 * torch.randn(N, generator=g))
    eps = 1e-6
    ref = F.linear(F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), eps), W.float(), bias.float())
    if act == 1:
        ref = F.gelu(ref)
    xd, Wd, gd, bd, biasd = (t.to(device).contiguous() for t in (x, W, gamma, beta, bias))
    Cd = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=device)
    stats = torch.empty(M, 2, dtype=torch.float32, device=device)
    ls, lc = torch.empty(N, dtype=torch.float32, device=device), torch.empty(N, dtype=torch.float32, device=device)
    L_.check(lib.emmax_op_gemm_ln(xd.data_ptr(), K, Wd.data_ptr(), K, Cd.data_ptr(), N, M, N, K, gd.data_ptr(), bd.data_ptr(), biasd.data_ptr(),
                                  eps, act, stats.data_ptr(), ls.data_ptr(), lc.data_ptr(), stream()), "gemm_ln")
    torch.cuda.synchronize()
    # the statistics pass alone
    mu, var = x.float().mean(-1), x.float().var(-1, unbiased=False)
    assert relerr(stats[:, 0], mu) < 1e-5 and relerr(stats[:, 1], (var + eps).rsqrt()) < 1e-5
    assert relerr(Cd, ref) < TOL, relerr(Cd, ref)
    # two rounded operands (x raw, W .* gamma) instead of one, and massive channels whose products are 40x a typical term: measured
    # worst 1.0e-2 x rms(ref) beyond rtol on 3 of 34 M elements (round 4) -- the same 2^-9 relative error on the dominant term that
    # rounding LN(x) to bf16 would leave
    assert_elementwise(Cd, ref, atol_frac=1.6e-2, what="gemm + folded LayerNorm")


@pytest.mark.parametrize("M,N,K,act", [(256, 256, 64, 0), (1000, 3456, 1152, 0), (2048, 1024, 4096, 1), (768, 4096, 1024, 2), (5000, 512, 192, 0)])
def test_gemm_main_loop_variants_are_bit_identical(device, tune, M, N, K, act):
    """The three main loops of the GEMM -- two LDS stages + one barrier per K step (rounds 1-3, tuning switch gemm_deep = 0), the deep A
    ring (1) and the staggered wave groups with counted waits across bare barriers (3; the default -1 picks by geometry) -- accumulate a
    tile's K steps in the same order: their outputs must agree BIT FOR BIT, on every repetition.  A DMA / fragment-read race in the
    counted-wait schedules shows up as a rare differing tile (tools/gemm_race_screen.py is the long form: 30 repetitions, 12 shapes)."""
    L_, lib = _lib()
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = bf(torch.randn(M, K, generator=g) * 0.5).to(device)
    W = bf(torch.randn(N, K, generator=g) * 0.05).to(device)
    bias = bf(torch.randn(N, generator=g) * 0.1).to(device) if act != 2 else None
    No = N // 2 if act == 2 else N
    outs = {}
    for deep in (0, -1, 1, 3):
        tune(gemm_deep=deep)
        for rep in range(4):
            out = torch.full((M, No), float("nan"), dtype=torch.bfloat16, device=device)
            L_.check(lib.emmax_op_gemm(A.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), No, M, N, K, L_.ptr(bias), act, None, None, 0, 0, stream()), "gemm")
            torch.cuda.synchronize()
            if deep == 0 and rep == 0:
                outs[0] = out
                assert torch.isfinite(out.float()).all()
            else:
                assert torch.equal(out.view(torch.int16), outs[0].view(torch.int16)), (deep, rep)
    # round 5: the 128 x 256 x 32 tile (two blocks per CU, three 32-deep stages, one counted-wait barrier per step; tuning switch
    # gemm_big = 2) accumulates a tile's K in the same order too -- against the 256 x 256 tile forced the same way (no split-K on either)
    tune(gemm_deep=-1, gemm_big=1)
    ref = torch.full((M, No), float("nan"), dtype=torch.bfloat16, device=device)
    L_.check(lib.emmax_op_gemm(A.data_ptr(), K, W.data_ptr(), K, ref.data_ptr(), No, M, N, K, L_.ptr(bias), act, None, None, 0, 0, stream()), "gemm")
    tune(gemm_big=2)
    for rep in range(6):
        out = torch.full((M, No), float("nan"), dtype=torch.bfloat16, device=device)
        L_.check(lib.emmax_op_gemm(A.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), No, M, N, K, L_.ptr(bias), act, None, None, 0, 0, stream()), "gemm k32")
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), ("k32", rep, (out.float() - ref.float()).abs().max().item())


def test_gemm_rejects_bad_shapes(device):
    L, lib = _lib()
    x = torch.zeros(128, 128, dtype=torch.bfloat16, device=device)
    assert lib.emmax_op_gemm(x.data_ptr(), 100, x.data_ptr(), 100, x.data_ptr(), 128, 128, 128, 100, None, 0, None, None, 0, 0, stream()) != 0
    assert b"emmax_op_gemm" in lib.emmax_last_error()


@pytest.mark.parametrize("rows,D", [(5, 128), (261, 1024), (256, 1152), (7, 144), (3, 4096)])
def test_layernorm_rmsnorm(device, rows, D):
    L, lib = _lib()
    g = torch.Generator().manual_seed(rows + D)
    x = bf(torch.randn(rows, D, generator=g) * 3 + 0.5)
    w = bf(1 + 0.1 * torch.randn(D, generator=g))
    b = bf(0.1 * torch.randn(D, generator=g))
    xd, wd, bd = x.to(device), w.to(device), b.to(device)
    y = torch.empty_like(xd)
    L.check(lib.emmax_op_layernorm(xd.data_ptr(), y.data_ptr(), wd.data_ptr(), bd.data_ptr(), rows, D, 1e-6, stream()), "ln")
    ref = F.layer_norm(x.float(), (D,), w.float(), b.float(), eps=1e-6)
    assert relerr(y, ref) < TOL
    assert_elementwise(y, ref)
    from oracle import emmax_oracle as orc

    L.check(lib.emmax_op_rmsnorm(xd.data_ptr(), y.data_ptr(), wd.data_ptr(), rows, D, 1e-5, stream()), "rms")
    assert relerr(y, orc.rms_norm(x.float(), w.float(), 1e-5)) < TOL


def _attn_ref(qkv, cu, Hq, Hkv, hd, q_off, k_off, v_off, scale, causal):
    outs = []
    for b in range(len(cu) - 1):
        x = qkv[cu[b]:cu[b + 1]].float()
        n = x.shape[0]
        q = x[:, q_off:q_off + Hq * hd].view(n, Hq, hd).transpose(0, 1)
        k = x[:, k_off:k_off + Hkv * hd].view(n, Hkv, hd).transpose(0, 1).repeat_interleave(Hq // Hkv, 0)
        v = x[:, v_off:v_off + Hkv * hd].view(n, Hkv, hd).transpose(0, 1).repeat_interleave(Hq // Hkv, 0)
        s = (q @ k.transpose(1, 2)) * scale
        if causal:
            s = s.masked_fill(torch.triu(torch.ones(n, n, dtype=torch.bool), 1), float("-inf"))
        outs.append((s.softmax(-1) @ v).transpose(0, 1).reshape(n, Hq * hd))
    return torch.cat(outs)


@pytest.mark.parametrize("hd,Hq,Hkv,lens,causal", [
    (64, 2, 2, [261], 0), (72, 2, 2, [256, 256], 0), (64, 16, 16, [261, 261], 0), (72, 16, 16, [256], 0),
    (128, 2, 2, [300], 1), (128, 4, 2, [70, 1, 129, 64], 1), (128, 2, 2, [768], 1), (128, 2, 1, [5, 200], 1),
])
def test_attention(device, hd, Hq, Hkv, lens, causal):
    _attention_case(device, hd, Hq, Hkv, lens, causal)


@pytest.mark.parametrize("ksplit", [0, 1])
@pytest.mark.parametrize("Hq,Hkv,lens", [(2, 2, [300]), (4, 2, [70, 1, 129, 64]), (2, 2, [768]), (2, 1, [5, 200]), (32, 32, [768]), (8, 8, [33, 1023])])
def test_attention_causal_with_and_without_key_groups(device, Hq, Hkv, lens, ksplit):
    """The causal head_dim-128 launch in both block forms (tuning switch attn_ksplit): four query waves per block, and eight waves as two
    key groups over the same four query waves whose (m, l, O) states meet through LDS -- the form an under-filled launch (one frame's
    prefill: 192 blocks) takes by default.  Includes the first 32 queries of a sequence, whose second key group sees no key at all."""
    L, _ = _lib()
    with L.tuning(attn_ksplit=ksplit):
        _attention_case(device, 128, Hq, Hkv, lens, 1)


def _attention_case(device, hd, Hq, Hkv, lens, causal):
    L, lib = _lib()
    g = torch.Generator().manual_seed(hd + sum(lens))
    total = sum(lens)
    qd, kvd = Hq * hd, Hkv * hd
    ld = (qd + 2 * kvd + 127) // 128 * 128
    qkv = bf(torch.randn(total, ld, generator=g))
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + n)
    scale = hd ** -0.5
    ref = _attn_ref(qkv, cu, Hq, Hkv, hd, 0, qd, qd + kvd, scale, causal)
    qkv_d = qkv.to(device)
    cu_d = torch.tensor(cu, dtype=torch.int32, device=device)
    ld_out = (qd + 127) // 128 * 128
    out = torch.zeros(total, ld_out, dtype=torch.bfloat16, device=device)
    L.check(lib.emmax_op_attention(qkv_d.data_ptr(), ld, 0, qd, qd + kvd, out.data_ptr(), ld_out, cu_d.data_ptr(), len(lens), max(lens),
                                   Hq, Hkv, hd, scale, causal, stream()), "attention")
    torch.cuda.synchronize()
    got = out[:, :qd]
    assert torch.isfinite(got.float()).all()
    assert relerr(got, ref) < TOL
    assert_elementwise(got, ref, atol_frac=ATTN_ATOL)
    if ld_out > qd:
        assert (out[:, qd:] == 0).all(), "attention must not write the padding columns"


@pytest.mark.parametrize("B", [1, 2, 3, 5, 8])
# (4096, 11008): the down-projection shape; (999, 8712): odd row count, K not a multiple of the 4096-element step
@pytest.mark.parametrize("N,K", [(256, 256), (4096, 4096), (1000, 11008), (64, 688), (4096, 11008), (999, 8712)])
def test_gemv(device, B, N, K):
    L, lib = _lib()
    g = torch.Generator().manual_seed(B + N + K)
    x = bf(torch.randn(B, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    ref = x.float() @ W.float().t()
    xd, Wd = x.to(device), W.to(device)
    y = torch.zeros(B, N, dtype=torch.bfloat16, device=device)
    # a session is what normally raises the dynamic-LDS limit; do it here through a throw-away tiny engine
    L.check(lib.emmax_op_gemv(xd.data_ptr(), Wd.data_ptr(), y.data_ptr(), B, N, K, stream()), "gemv")
    torch.cuda.synchronize()
    assert relerr(y, ref) < TOL
    assert_elementwise(y, ref)


@pytest.mark.parametrize("B", [1, 3, 4, 8])
@pytest.mark.parametrize("N,K", [(256, 256), (4096, 4096), (1008, 11008), (64, 704), (32064, 256)])
def test_gemm_small_mfma(device, B, N, K):
    """Batch >= 3 decode projection: MFMA over the fragment-major weight copy (also exercised at B = 1 for the layout)."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(B * 3 + N + K)
    x = bf(torch.randn(B, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    ref = x.float() @ W.float().t()
    xd, Wd = x.to(device), W.to(device)
    Wfm = torch.empty_like(Wd)
    L.check(lib.emmax_op_repack_fm(Wd.data_ptr(), K, Wfm.data_ptr(), N, K, stream()), "repack")
    # layout spot check: tile (nt, kt), lane -> W[16 nt + lane%16][32 kt + 8 (lane//16) : +8]
    flat = Wfm.view(-1, 64, 8).cpu()
    nt, kt, lane = (N // 16) - 1, (K // 32) - 1, 37
    assert torch.equal(flat[nt * (K // 32) + kt, lane], W[16 * nt + lane % 16, 32 * kt + 8 * (lane // 16): 32 * kt + 8 * (lane // 16) + 8])
    y = torch.full((B, N), float("nan"), dtype=torch.bfloat16, device=device)
    L.check(lib.emmax_op_gemm_small(xd.data_ptr(), Wfm.data_ptr(), y.data_ptr(), B, N, K, stream()), "gemm_small")
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all()
    assert relerr(y, ref) < TOL
    assert_elementwise(y, ref)


@pytest.mark.parametrize("B", [17, 24, 32])
@pytest.mark.parametrize("N,K", [(256, 256), (4096, 4096), (12288, 4096), (22016, 4096), (32064, 4096), (1008, 1024), (48, 2048), (4096, 11008), (64, 8192),
                                 (48, 11264), (512, 3072), (64, 320), (32, 11040)])
def test_gemm_small_kmp(device, B, N, K):
    """Batch 17-32 decode projection (decode_kmp.hip, round 5): two 16-wide batch tiles per weight tile, the wave's K slice in phases of
    four k-steps through a 32-row LDS window, accumulators of all tiles of the block in registers.  N covers the three block shapes
    (1, <= 3, <= 6 tiles per block; 32064: more blocks than CUs), K the 4-phase form (<= 4096, incl. slices shorter than four phases)
    and the 11-phase form of the down projection (K = 11008, 8192, 11264 = its limit); K = 320 / 11040: wave slices in whole k-steps
    that differ by one (10 / 345 steps over eight waves; 11040: the widest slice fills the eleven phases)."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(B * 13 + N + K)
    x = bf(torch.randn(B, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    ref = x.float() @ W.float().t()
    xd, Wd = x.to(device), W.to(device)
    Wk = torch.empty_like(Wd)
    y = torch.full((B, N), float("nan"), dtype=torch.bfloat16, device=device)
    L.check(lib.emmax_op_repack_km(Wd.data_ptr(), K, Wk.data_ptr(), N, K, 0, 0, stream()), "repack km")
    L.check(lib.emmax_op_gemm_small_km(xd.data_ptr(), Wk.data_ptr(), y.data_ptr(), B, N, K, stream()), "gemm small kmp")
    torch.cuda.synchronize()
    assert relerr(y, ref) < TOL
    assert_elementwise(y, ref)


@pytest.mark.parametrize("B", [3, 5, 8, 9, 13, 16])
@pytest.mark.parametrize("N,K", [(256, 256), (4096, 4096), (12288, 4096), (32064, 4096), (1008, 1024), (48, 2048), (4096, 11008), (64, 8192), (64, 4352),
                                 (48, 11264)])
def test_gemm_small_km(device, B, N, K):
    """Batch 3-16 decode projection on the K-split MFMA kernel (decode_km.hip): activations as register fragments, two tiles in
    flight per wave, partial tiles of the eight K slices meeting in LDS.  N covers 1 .. 8 tiles per block and ragged shares.
    Round 5: batches 9-16 (the MFMA's N side is 16 wide: sixteen staged rows per wave, window and partial tiles sharing one LDS
    region) and, for K > 4096, the phased kernel of the down projection (two phases of 22 fragments at B <= 8, four of 12 above;
    K = 11264 = its limit at B <= 8, K = 4352: a slice shorter than one phase at B <= 8)."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(B * 11 + N + K)
    x = bf(torch.randn(B, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    ref = x.float() @ W.float().t()
    xd, Wd = x.to(device), W.to(device)
    Wk = torch.empty_like(Wd)
    y = torch.full((B, N), float("nan"), dtype=torch.bfloat16, device=device)
    L.check(lib.emmax_op_repack_km(Wd.data_ptr(), K, Wk.data_ptr(), N, K, 0, 0, stream()), "repack km")
    L.check(lib.emmax_op_gemm_small_km(xd.data_ptr(), Wk.data_ptr(), y.data_ptr(), B, N, K, stream()), "gemm small km")
    torch.cuda.synchronize()
    assert relerr(y, ref) < TOL
    assert_elementwise(y, ref)


def test_gemm_small_km_refuses_shapes_outside_it(device):
    L, lib = _lib()
    x = torch.zeros(33, 12320, dtype=torch.bfloat16, device=device)
    W = torch.zeros(8192 * 12320, dtype=torch.bfloat16, device=device)
    y = torch.zeros(33, 8192, dtype=torch.bfloat16, device=device)
    for B, N, K in [(4, 64, 11296), (4, 64, 320), (33, 64, 256), (4, 40, 256), (16, 64, 12320), (20, 64, 11520), (20, 8192, 11008), (20, 64, 336), (20, 64, 224), (20, 64, 11296)]:
        assert lib.emmax_op_gemm_small_km(x.data_ptr(), W.data_ptr(), y.data_ptr(), B, N, K, stream()) != 0, (B, N, K)
    torch.cuda.synchronize()


@pytest.mark.parametrize("H,W", [(256, 256), (480, 640), (224, 300), (100, 180), (500, 224)])
def test_device_resize_matches_pillow(device, H, W):
    """Device-side resize-naive == PIL.Image.resize((224,224), BICUBIC) bit for bit (uint8)."""
    from PIL import Image

    from emmax.config import EmmaXConfig
    from emmax.resize import bicubic_coeffs

    L, lib = _lib()
    rng = np.random.default_rng(H * 7 + W)
    frames = rng.integers(0, 256, size=(2, H, W, 3), dtype=np.uint8)
    ref = np.stack([np.asarray(Image.fromarray(f).resize((224, 224), Image.BICUBIC)) for f in frames])
    src = torch.from_numpy(frames).to(device)
    dst = torch.zeros(2, 224, 224, 3, dtype=torch.uint8, device=device)
    tmp = torch.zeros(2, H, 224, 3, dtype=torch.uint8, device=device)
    bh, kh, nh = bicubic_coeffs(W, 224)
    bv, kv, nv = bicubic_coeffs(H, 224)
    t = [torch.from_numpy(a.copy()).to(device) for a in (bh, kh, bv, kv)]
    L.check(lib.emmax_op_resize_bicubic_u8(src.data_ptr(), 2, H, W, dst.data_ptr(), 224, 224, tmp.data_ptr(), t[0].data_ptr(), t[1].data_ptr(), nh,
                                           t[2].data_ptr(), t[3].data_ptr(), nv, stream()), "resize")
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), ref)


@pytest.mark.parametrize("B", [1, 4, 8])
@pytest.mark.parametrize("N,K", [(256, 256), (4096, 4096), (1008, 11008), (64, 704)])
def test_fp8_weight_projection(device, B, N, K):
    """fp8-e4m3 weight copy (per-row scale, fragment-major) de-quantised in registers: the device quantiser must agree
    with torch.float8_e4m3fn (OCP) bit for bit, and the projection with an fp32 matmul over the de-quantised weights."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(B * 5 + N + K)
    x = bf(torch.randn(B, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    W[3, 5] = 0.7   # an outlier that sets the row scale
    scale_ref = W.float().abs().amax(dim=1).clamp_min(1e-30) / 448.0
    Wq_ref = (W.float() / scale_ref[:, None]).to(torch.float8_e4m3fn)
    Wd, xd = W.to(device), x.to(device)
    W8 = torch.empty(N * K, dtype=torch.uint8, device=device)
    sc = torch.empty(N, dtype=torch.float32, device=device)
    L.check(lib.emmax_op_quant_fm8(Wd.data_ptr(), K, W8.data_ptr(), sc.data_ptr(), N, K, stream()), "quant")
    torch.cuda.synchronize()
    assert torch.allclose(sc.cpu(), scale_ref, rtol=1e-6, atol=0)
    # undo the fragment-major layout: tile (nt, kt2) lane (g, i) bytes [0:8] = k 8g.., [8:16] = k 32+8g..
    t = W8.cpu().view(N // 16, K // 64, 4, 16, 2, 8)            # nt, kt2, g, i, half, j
    back = t.permute(0, 3, 1, 4, 2, 5).reshape(N, K)              # (nt, i) x (kt2, half, g, j)
    diff = (back != Wq_ref.view(torch.uint8))
    assert not diff.any(), f"{int(diff.sum())} of {diff.numel()} fp8 codes differ from torch.float8_e4m3fn"
    ref = x.float() @ (Wq_ref.float() * scale_ref[:, None]).t()
    y = torch.full((B, N), float("nan"), dtype=torch.bfloat16, device=device)
    L.check(lib.emmax_op_gemm_small_fp8(xd.data_ptr(), W8.data_ptr(), sc.data_ptr(), y.data_ptr(), B, N, K, stream()), "gemm fp8")
    torch.cuda.synchronize()
    assert relerr(y, ref) < TOL
    assert_elementwise(y, ref)


@pytest.mark.parametrize("B", [9, 16, 17, 24, 32])
@pytest.mark.parametrize("N,K", [(4096, 4096), (1008, 11008), (256, 512), (64, 704)])
def test_fp8_weight_projection_nine_to_thirty_two_rows(device, B, N, K):
    """The same e4m3 tiles on the K-split kernels: 9-16 rows decode_km.hip (K <= 4096; beyond, its phased down form), 17-32 rows
    decode_kmp.hip -- two fragments per 1 KiB load step, wave slices in whole load steps that may differ by one (K = 11008: 172
    steps over eight waves; K = 704: 11).  Against an fp32 matmul over the de-quantised weights."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(B * 7 + N + K)
    x = bf(torch.randn(B, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    W[5, 3] = 0.6
    scale_ref = W.float().abs().amax(dim=1).clamp_min(1e-30) / 448.0
    Wq_ref = (W.float() / scale_ref[:, None]).to(torch.float8_e4m3fn)
    Wd, xd = W.to(device), x.to(device)
    W8 = torch.empty(N * K, dtype=torch.uint8, device=device)
    sc = torch.empty(N, dtype=torch.float32, device=device)
    L.check(lib.emmax_op_quant_fm8(Wd.data_ptr(), K, W8.data_ptr(), sc.data_ptr(), N, K, stream()), "quant")
    y = torch.full((B, N), float("nan"), dtype=torch.bfloat16, device=device)
    rc = lib.emmax_op_gemm_small_fp8(xd.data_ptr(), W8.data_ptr(), sc.data_ptr(), y.data_ptr(), B, N, K, stream())
    if B <= 16 and K <= 4096 and K % 512:
        assert rc != 0 and b"emmax_op_gemm_small_fp8" in lib.emmax_last_error()   # decode_km.hip: equal wave slices (K % 512) below the phased form
        return
    L.check(rc, "gemm fp8")
    torch.cuda.synchronize()
    ref = x.float() @ (Wq_ref.float() * scale_ref[:, None]).t()
    assert relerr(y, ref) < TOL
    assert_elementwise(y, ref)


def _rm8_rows_to_codes(W8, N, K):
    """Undo the span order of emmax_quant_rm8_kernel: spans of 128 chunks of 8 bytes (the last one shorter, nc chunks);
    granule l of a span = chunk l ++ chunk nc/2 + l."""
    rows = W8.cpu().view(N, K)
    back = torch.empty_like(rows)
    nch = K // 8
    for sp in range((nch + 127) // 128):
        nc = min(128, nch - sp * 128)
        span = rows[:, sp * 1024: sp * 1024 + nc * 8].reshape(N, nc // 2, 2, 8)      # granule, half, byte
        back[:, sp * 1024: sp * 1024 + nc * 8] = span.permute(0, 2, 1, 3).reshape(N, nc * 8)
    return back


@pytest.mark.parametrize("B", [1, 2])
@pytest.mark.parametrize("N,K", [(256, 256), (4096, 4096), (1008, 11008), (64, 704), (7, 64), (33, 1040), (4099, 2048),
                                 (12290, 1040), (22016, 4096), (12293, 64)])
def test_fp8_row_gemv(device, B, N, K):
    """Batch 1-2 of the fp8 decode weights: the e4m3 ROW copy (span order, decode.hip) must hold torch.float8_e4m3fn's codes
    bit for bit with the scales of the fragment-major quantiser, and the dot-product GEMV over it must match an fp32 matmul
    over the de-quantised weights -- all three block shapes of the kernel (two rows x 4 or 8 steps, four-row groups from 12288
    rows), row counts that are not a multiple of the group, K with a short last span (704, 1040, 11008) and K below one span (64)."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(B * 7 + N + K)
    x = bf(torch.randn(B, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    W[3, 5] = 0.7
    scale_ref = W.float().abs().amax(dim=1).clamp_min(1e-30) / 448.0
    Wq_ref = (W.float() / scale_ref[:, None]).to(torch.float8_e4m3fn)
    Wd, xd = W.to(device), x.to(device)
    W8 = torch.empty(N * K, dtype=torch.uint8, device=device)
    sc = torch.empty(N, dtype=torch.float32, device=device)
    L.check(lib.emmax_op_quant_rm8(Wd.data_ptr(), K, W8.data_ptr(), sc.data_ptr(), N, K, stream()), "quant rows")
    torch.cuda.synchronize()
    assert torch.allclose(sc.cpu(), scale_ref, rtol=1e-6, atol=0)
    diff = (_rm8_rows_to_codes(W8, N, K) != Wq_ref.view(torch.uint8))
    assert not diff.any(), f"{int(diff.sum())} of {diff.numel()} fp8 codes differ from torch.float8_e4m3fn"
    ref = x.float() @ (Wq_ref.float() * scale_ref[:, None]).t()
    y = torch.full((B, N), float("nan"), dtype=torch.bfloat16, device=device)
    L.check(lib.emmax_op_gemv_fp8(xd.data_ptr(), W8.data_ptr(), sc.data_ptr(), y.data_ptr(), B, N, K, stream()), "gemv fp8")
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all()
    assert relerr(y, ref) < TOL
    assert_elementwise(y, ref)


def test_fp8_row_gemv_refuses_what_it_cannot_stage(device):
    """The fp8-row GEMV stages whole activation rows in LDS (one K phase) and serves batch 1-2: a third row, a K that is not a
    multiple of 16 or rows beyond the LDS budget are refused with an error code and a message (the session then keeps such a
    projection on the MFMA kernel, model.hip: launch_proj), never run wrong."""
    L, lib = _lib()
    x = torch.zeros(3, 65536, dtype=torch.bfloat16, device=device)
    W8 = torch.zeros(16 * 65536, dtype=torch.uint8, device=device)
    sc = torch.ones(16, dtype=torch.float32, device=device)
    y = torch.zeros(3, 16, dtype=torch.bfloat16, device=device)
    for B, N, K in [(3, 16, 256), (2, 16, 65536), (1, 16, 264)]:
        rc = lib.emmax_op_gemv_fp8(x.data_ptr(), W8.data_ptr(), sc.data_ptr(), y.data_ptr(), B, N, K, stream())
        assert rc != 0, (B, N, K)
        assert b"emmax_op_gemv_fp8" in lib.emmax_last_error()
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------------------------------
# Paged split-KV decode attention (emmax_decode_attn_kernel) at the benchmark's operating point: contexts 768..1280,
# ragged batches, MHA and GQA, every split count the launcher can pick, page tables that are NOT the identity.
# Reference = the oracle's cached attention (oracle.emmax_oracle.llama_layer: fp32 softmax over keys 0..ctx) on the same
# bf16-rounded q / K / V.
# ---------------------------------------------------------------------------------------------------------------------
def _paged_cache(K, V, page, max_pages, gen):
    """K, V: lists (per row) of [L_b, Hkv, 128] bf16 -> (kcache, vcache [n_pages][Hkv][page][128], page_table [B][max_pages])."""
    B = len(K)
    Hkv = K[0].shape[1]
    n_pages = B * max_pages
    perm = torch.randperm(n_pages, generator=gen)
    table = perm.view(B, max_pages).to(torch.int32)
    kc = torch.randn(n_pages, Hkv, page, 128, generator=gen).to(torch.bfloat16)   # garbage beyond the context must not matter
    vc = torch.randn(n_pages, Hkv, page, 128, generator=gen).to(torch.bfloat16)
    for b in range(B):
        L = K[b].shape[0]
        for t0 in range(0, L, page):
            pg = int(table[b, t0 // page])
            n = min(page, L - t0)
            kc[pg, :, :n] = K[b][t0:t0 + n].transpose(0, 1)
            vc[pg, :, :n] = V[b][t0:t0 + n].transpose(0, 1)
    return kc, vc, table


def _merge_partials(part, nsplit):
    """f32 [B,Hq,nsplit,132] -> normalised o [B,Hq,128] (the merge the o-proj prologue performs, in fp32)."""
    o, m, l = part[..., :128], part[..., 128], part[..., 129]
    M = m.max(dim=-1, keepdim=True).values
    w = torch.where(torch.isinf(m), torch.zeros_like(m), torch.exp(m - M))
    den = (l * w).sum(-1)
    return (o * w[..., None]).sum(-2) / den[..., None]


@pytest.mark.parametrize("Hq,Hkv", [(32, 32), (4, 2), (8, 1)])
@pytest.mark.parametrize("ctxs", [[63], [64], [65], [767], [768], [1023], [1024], [1025], [1279], [768, 63, 1279], [1279, 1, 64, 1024, 65, 767, 1025, 300]])
def test_decode_attention_paged_matches_oracle(device, Hq, Hkv, ctxs):
    L_, lib = _lib()
    B, page, max_pages = len(ctxs), 64, 21
    g = torch.Generator().manual_seed(Hq * 1000 + sum(ctxs))
    scale = 128 ** -0.5
    q = bf(torch.randn(B, Hq, 128, generator=g))
    K = [bf(torch.randn(c + 1, Hkv, 128, generator=g)) for c in ctxs]   # keys 0..ctx inclusive
    V = [bf(torch.randn(c + 1, Hkv, 128, generator=g)) for c in ctxs]
    kc, vc, table = _paged_cache(K, V, page, max_pages, g)
    rep = Hq // Hkv
    ref = torch.empty(B, Hq, 128)
    for b in range(B):
        kk = K[b].float().repeat_interleave(rep, dim=1)          # [L, Hq, 128]
        vv = V[b].float().repeat_interleave(rep, dim=1)
        att = torch.einsum("hd,lhd->hl", q[b].float(), kk) * scale
        ref[b] = torch.einsum("hl,lhd->hd", F.softmax(att, dim=-1, dtype=torch.float32), vv)
    qd, kcd, vcd, td = q.view(B, Hq * 128).contiguous().to(device), kc.to(device), vc.to(device), table.to(device)
    ctx_d = torch.tensor(ctxs, dtype=torch.int32, device=device)
    ns_auto = C.c_int()
    for nsplit in (0, 1, 2, 4, 8, 16):
        ns = nsplit
        part = torch.full((B, Hq, max(nsplit, 16), 132), float("nan"), dtype=torch.float32, device=device)
        L_.check(lib.emmax_op_decode_attention(qd.data_ptr(), kcd.data_ptr(), vcd.data_ptr(), td.data_ptr(), ctx_d.data_ptr(), None,
                                               part.data_ptr(), B, Hq, Hkv, page, max_pages, nsplit, scale, C.byref(ns_auto), stream()),
                 "decode attention")
        torch.cuda.synchronize()
        ns = ns_auto.value
        assert ns >= 1 and (ns & (ns - 1)) == 0
        got = _merge_partials(part.view(-1)[: B * Hq * ns * 132].view(B, Hq, ns, 132).cpu(), ns)
        assert torch.isfinite(got).all(), (nsplit, ns)
        assert relerr(got, ref) < 5e-3, (nsplit, ns, relerr(got, ref))   # inputs are exact bf16, math fp32: only exp/ordering noise
        assert_elementwise(got, ref)
    # the ONE-split direct form (what the decode step launches from batch 5 up at 32 heads): the block normalises its result and
    # writes the bf16 row itself -- compared with the bf16 rounding of the merged partials (same arithmetic, fp32 merge of one split)
    o = torch.full((B, Hq * 128), float("nan"), dtype=torch.bfloat16, device=device)
    L_.check(lib.emmax_op_decode_attention_direct(qd.data_ptr(), kcd.data_ptr(), vcd.data_ptr(), td.data_ptr(), ctx_d.data_ptr(), None,
                                                  o.data_ptr(), B, Hq, Hkv, page, max_pages, scale, stream()), "decode attention, direct")
    torch.cuda.synchronize()
    om = o.float().cpu().view(B, Hq, 128)
    assert torch.isfinite(om).all()
    assert relerr(om, ref) < 8e-3, relerr(om, ref)
    assert_elementwise(om, ref)


def _quant_rows_e4m3(x):
    """[..., 128] float -> (e4m3 bytes as uint8, fp32 scale per row = the smallest power of two with amax / scale <= 448, what the kernels
    read back: e4m3 x scale, exact in bf16)."""
    r = (x.abs().amax(dim=-1, keepdim=True).float() / 448.0)
    sc = torch.where(r > 0, torch.exp2(torch.ceil(torch.log2(r.clamp_min(1e-38)))), torch.ones_like(r)).float()   # power of two (common.h: e4m3_row_scale)
    q8 = (x.float() / sc).to(torch.float8_e4m3fn)
    deq = (q8.float() * sc).to(torch.bfloat16).float()
    return q8.view(torch.uint8), sc.squeeze(-1), deq


@pytest.mark.parametrize("Hq,Hkv", [(32, 32), (8, 2)])
@pytest.mark.parametrize("ctxs", [[63], [64], [767], [1024], [1279], [768, 63, 1279], [1279, 1, 64, 1024, 65, 767, 1025, 300, 5, 900, 64, 128, 333, 1000, 2, 640]])
def test_decode_attention_over_the_fp8_kv_cache(device, Hq, Hkv, ctxs):
    """The opt-in fp8 KV cache (round 5, tuning switch kv_fp8): e4m3 K / V rows with one fp32 scale per (token, head) row.  The kernel
    must (a) attend over bf16(e4m3 x scale) of the cached keys and of the step's NEW key (handed over as bf16 in the staging rows),
    checked against an fp32 softmax over exactly those values, every split count and the direct form; (b) append the new key's bytes
    and scale at position ctx_len -- checked against torch's own e4m3 rounding of the staged rows."""
    L_, lib = _lib()
    B, page, max_pages = len(ctxs), 64, 21
    g = torch.Generator().manual_seed(Hq * 77 + sum(ctxs))
    scale = 128 ** -0.5
    q = bf(torch.randn(B, Hq, 128, generator=g))
    K = [bf(torch.randn(c + 1, Hkv, 128, generator=g) * (0.5 + 2.0 * torch.rand(c + 1, Hkv, 1, generator=g))) for c in ctxs]   # keys 0..ctx; row scales vary
    V = [bf(torch.randn(c + 1, Hkv, 128, generator=g) * (0.5 + 2.0 * torch.rand(c + 1, Hkv, 1, generator=g))) for c in ctxs]
    K8, KS, KD = zip(*[_quant_rows_e4m3(k) for k in K])
    V8, VS, VD = zip(*[_quant_rows_e4m3(v) for v in V])
    rep = Hq // Hkv
    ref = torch.empty(B, Hq, 128)
    for b in range(B):
        kk, vv = KD[b].repeat_interleave(rep, dim=1), VD[b].repeat_interleave(rep, dim=1)
        att = torch.einsum("hd,lhd->hl", q[b].float(), kk) * scale
        ref[b] = torch.einsum("hl,lhd->hd", F.softmax(att, dim=-1, dtype=torch.float32), vv)
    # paged byte caches + scales holding keys 0 .. ctx - 1 (the new key, position ctx, arrives through the staging rows)
    n_pages = B * max_pages
    table = torch.randperm(n_pages, generator=g).view(B, max_pages).to(torch.int32)
    kc = torch.randint(0, 120, (n_pages, Hkv, page, 128), generator=g, dtype=torch.uint8)     # garbage beyond the context must not matter
    vc = torch.randint(0, 120, (n_pages, Hkv, page, 128), generator=g, dtype=torch.uint8)
    ks, vs = torch.rand(n_pages, Hkv, page, generator=g), torch.rand(n_pages, Hkv, page, generator=g)
    for b, c in enumerate(ctxs):
        for t0 in range(0, c, page):
            pg, n = int(table[b, t0 // page]), min(page, c - t0)
            kc[pg, :, :n], vc[pg, :, :n] = K8[b][t0:t0 + n].transpose(0, 1), V8[b][t0:t0 + n].transpose(0, 1)
            ks[pg, :, :n], vs[pg, :, :n] = KS[b][t0:t0 + n].transpose(0, 1), VS[b][t0:t0 + n].transpose(0, 1)
    stage = torch.stack([torch.stack([K[b][ctxs[b]], V[b][ctxs[b]]], dim=1) for b in range(B)]).contiguous()   # [B][Hkv][2][128] bf16
    qd, td, sd = q.view(B, Hq * 128).contiguous().to(device), table.to(device), stage.to(device)
    ctx_d = torch.tensor(ctxs, dtype=torch.int32, device=device)
    for nsplit in (1, 2, 8, 0):
        kcd, vcd, ksd, vsd = kc.to(device), vc.to(device), ks.to(device), vs.to(device)
        direct = nsplit == 0
        part = torch.full((B, Hq, 16, 132), float("nan"), dtype=torch.float32, device=device)
        o = torch.full((B, Hq * 128), float("nan"), dtype=torch.bfloat16, device=device)
        L_.check(lib.emmax_op_decode_attention_kv8(qd.data_ptr(), kcd.data_ptr(), vcd.data_ptr(), ksd.data_ptr(), vsd.data_ptr(), sd.data_ptr(),
                                                   td.data_ptr(), ctx_d.data_ptr(), None, None if direct else part.data_ptr(),
                                                   o.data_ptr() if direct else None, B, Hq, Hkv, page, max_pages, max(nsplit, 1), scale, stream()),
                 "decode attention kv8")
        torch.cuda.synchronize()
        got = o.float().cpu().view(B, Hq, 128) if direct else _merge_partials(part.view(-1)[: B * Hq * nsplit * 132].view(B, Hq, nsplit, 132).cpu(), nsplit)
        assert torch.isfinite(got).all(), nsplit
        assert relerr(got, ref) < (8e-3 if direct else 5e-3), (nsplit, relerr(got, ref))
        # the appended row: bytes and scale at position ctx of every (row, head)
        kc2, vc2, ks2, vs2 = kcd.cpu(), vcd.cpu(), ksd.cpu(), vsd.cpu()
        for b, c in enumerate(ctxs):
            pg, sl = int(table[b, c // page]), c % page
            for got8, gots, want8, wants, wantd in ((kc2, ks2, K8[b][c], KS[b][c], KD[b][c]), (vc2, vs2, V8[b][c], VS[b][c], VD[b][c])):
                assert torch.equal(gots[pg, :, sl], wants), (b, "scale")
                assert torch.equal(got8[pg, :, sl], want8), (b, "bytes")     # power-of-two scale: both roundings are the same exact operation


def test_decode_attention_done_rows_read_nothing(device):
    """Rows flagged done (finished / idle slots) must produce empty partials (m = -inf, l = 0) and leave the others untouched."""
    L_, lib = _lib()
    B, Hq, Hkv, page, max_pages = 3, 32, 32, 64, 21
    ctxs = [900, 500, 1100]
    g = torch.Generator().manual_seed(77)
    q = bf(torch.randn(B, Hq * 128, generator=g))
    K = [bf(torch.randn(c + 1, Hkv, 128, generator=g)) for c in ctxs]
    V = [bf(torch.randn(c + 1, Hkv, 128, generator=g)) for c in ctxs]
    kc, vc, table = _paged_cache(K, V, page, max_pages, g)
    dev = lambda t: t.to(device)
    qd, kcd, vcd, td = dev(q), dev(kc), dev(vc), dev(table)
    ctx_d = torch.tensor(ctxs, dtype=torch.int32, device=device)
    outs = []
    for done in ([0, 0, 0], [0, 1, 0]):
        done_d = torch.tensor(done, dtype=torch.int32, device=device)
        part = torch.full((B, Hq, 4, 132), float("nan"), dtype=torch.float32, device=device)
        L_.check(lib.emmax_op_decode_attention(qd.data_ptr(), kcd.data_ptr(), vcd.data_ptr(), td.data_ptr(), ctx_d.data_ptr(),
                                               done_d.data_ptr(), part.data_ptr(), B, Hq, Hkv, page, max_pages, 4, 128 ** -0.5, None, stream()),
                 "decode attention")
        torch.cuda.synchronize()
        outs.append(part.cpu())
    a, b_ = outs
    assert torch.equal(a[0], b_[0]) and torch.equal(a[2], b_[2])
    assert torch.isinf(b_[1, :, :, 128]).all() and (b_[1, :, :, 128] < 0).all() and (b_[1, :, :, 129] == 0).all() and (b_[1, :, :, :128] == 0).all()


@pytest.mark.parametrize("hd,Hq,Hkv,top,boost", [(64, 16, 16, 261, 1), (64, 16, 16, 288, 1), (64, 16, 16, 256, 8), (72, 16, 16, 256, 1), (64, 32, 8, 200, 1),
                                                 (72, 16, 4, 130, 8), (64, 16, 16, 261, 8)])
def test_attention_resident_form_many_short_sequences(device, hd, Hq, Hkv, top, boost):
    """>= 512 (head_dim 64) / 2048 (head_dim 72) (sequence, head) items of <= 288 / 256 tokens, non-causal, take the resident kernel
    (whole K / V of an item in LDS, double buffered, producer waves, swizzled / unpadded images): ragged lengths around every
    32-key step boundary, the production lengths 261 / 256, the maximum, single tokens, and grouped K / V heads.  boost = 8
    scales the queries so that the row maxima keep moving by more than the lazy-rescale threshold in later steps."""
    L, lib = _lib()
    rng = np.random.default_rng(hd * 1000 + top + Hkv)
    B = (512 if hd == 64 else 2048) // Hq + 3
    lens = [top, 1, 31, 0, 32, 33, 63, 64, 65, top - 1, max(1, top - 31), max(1, top - 32), 97, 160, 5, 0][: B]   # incl. empty sequences
    lens += [int(x) for x in rng.integers(1, top + 1, size=B - len(lens))]
    g = torch.Generator().manual_seed(hd + top)
    total = sum(lens)
    qd, kvd = Hq * hd, Hkv * hd
    ld = (qd + 2 * kvd + 127) // 128 * 128
    qkv = torch.randn(total, ld, generator=g)
    qkv[:, :qd] *= boost
    qkv = bf(qkv)
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + n)
    scale = hd ** -0.5
    ref = _attn_ref(qkv, cu, Hq, Hkv, hd, 0, qd, qd + kvd, scale, 0)
    ld_out = (qd + 127) // 128 * 128
    out = torch.zeros(total, ld_out, dtype=torch.bfloat16, device=device)
    cu_d = torch.tensor(cu, dtype=torch.int32, device=device)
    qkv_d = qkv.to(device)
    L.check(lib.emmax_op_attention(qkv_d.data_ptr(), ld, 0, qd, qd + kvd, out.data_ptr(), ld_out, cu_d.data_ptr(), len(lens), max(lens),
                                   Hq, Hkv, hd, scale, 0, stream()), "attention")
    torch.cuda.synchronize()
    got = out[:, :qd].cpu()
    assert torch.isfinite(got.float()).all()
    assert relerr(got, ref) < TOL
    assert_elementwise(got, ref, atol_frac=ATTN_ATOL)
    for b in (0, 1, 2, 4, len(lens) - 1):   # per sequence too: a wrong row of a short one disappears in the global norm
        assert relerr(got[cu[b]:cu[b + 1]], ref[cu[b]:cu[b + 1]]) < 2 * TOL, (b, lens[b])
    if ld_out > qd:
        assert (out[:, qd:] == 0).all(), "attention must not write the padding columns"


@pytest.mark.parametrize("seed", range(12))
def test_attention_random_ragged_shapes(device, seed):
    L, _ = _lib()
    with L.tuning(attn_ksplit=seed // 3 % 2 if seed % 3 == 2 else -1):     # head_dim 128: both block forms
        _attention_random_ragged(device, seed)


def _attention_random_ragged(device, seed):
    """Seeded random ragged batches through the attention kernel (the LDS-DMA ring, the masked / 32-key tail steps, the
    3- and 4-wave block shapes and the XCD work map all depend on the lengths): 1..6 sequences of 1..900 tokens, each head
    size, causal for head_dim 128, against the fp32 reference."""
    L, lib = _lib()
    rng = np.random.default_rng(1000 + seed)
    hd = [64, 72, 128][seed % 3]
    causal = 1 if hd == 128 else int(rng.integers(0, 2))
    Hkv = int(rng.choice([1, 2, 4]))
    Hq = Hkv * (int(rng.choice([1, 2])) if hd == 128 else 1)
    lens = [int(x) for x in rng.integers(1, 900, size=int(rng.integers(1, 7)))]
    if seed % 4 == 0:
        lens[0] = [261, 256, 768, 97][seed // 4 % 4]        # the production lengths and an odd one
    g = torch.Generator().manual_seed(seed)
    total = sum(lens)
    qd, kvd = Hq * hd, Hkv * hd
    ld = (qd + 2 * kvd + 127) // 128 * 128
    qkv = bf(torch.randn(total, ld, generator=g))
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + n)
    scale = hd ** -0.5
    ref = _attn_ref(qkv, cu, Hq, Hkv, hd, 0, qd, qd + kvd, scale, causal)
    qkv_d = qkv.to(device)
    cu_d = torch.tensor(cu, dtype=torch.int32, device=device)
    ld_out = (qd + 127) // 128 * 128
    out = torch.zeros(total, ld_out, dtype=torch.bfloat16, device=device)
    for _ in range(3):      # repeated launches: a ring / barrier race would not be deterministic
        L.check(lib.emmax_op_attention(qkv_d.data_ptr(), ld, 0, qd, qd + kvd, out.data_ptr(), ld_out, cu_d.data_ptr(), len(lens), max(lens),
                                       Hq, Hkv, hd, scale, causal, stream()), "attention")
        torch.cuda.synchronize()
        got = out[:, :qd]
        assert torch.isfinite(got.float()).all()
        assert relerr(got, ref) < TOL, (hd, Hq, Hkv, lens, causal, relerr(got, ref))
        assert_elementwise(got, ref, atol_frac=ATTN_ATOL)
        # per-row check as well: a wrong row with small values hides under the max-norm
        err_row = (got.float().cpu() - ref).abs().amax(dim=1) / ref.abs().amax(dim=1).clamp_min(1e-3)
        assert err_row.max().item() < 5e-2, (hd, lens, causal, err_row.max().item())
