"""CPU: the C-ABI shared library loads without a GPU, exports every symbol include/emmax.h declares, and its host-only
entry points (config validation, arena / session sizing, error strings) behave.  No compute calls."""

import ctypes as C
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from emmax import _lib

    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return _lib, _lib.load()


def test_every_declared_symbol_is_exported(lib):
    L, so = lib
    header = open(os.path.join(ROOT, "include", "emmax.h")).read()
    declared = set(re.findall(r"\b(emmax_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(so, name), f"{name} declared in include/emmax.h but not exported"
    assert declared == set(L.SIGNATURES), "ctypes signature table and header drifted apart"
    assert so.emmax_abi_version() == L.ABI_VERSION and b"gfx950" in so.emmax_version()
    m = re.search(r"#define EMMAX_ABI_VERSION (\d+)", header)
    assert m and int(m.group(1)) == L.ABI_VERSION


def _c_struct_fields(header, name):
    """(type, field) pairs of `typedef struct name { ... } name;` in the header, comments stripped, arrays as `field[n]`."""
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ty, rest = decl.split(None, 1)
        out += [(ty, f.strip()) for f in rest.split(",")]
    return out


def test_config_struct_layout_header_binding_and_integration_stub_agree(lib):
    """INTEGRATION.md's reference-side stub had drifted from emmax_config (no `decode_fp8`: a struct 4 bytes short).
    The header's struct, the ctypes mirror in emmax/_lib.py, the stub printed in INTEGRATION.md and the library's own
    sizeof(emmax_config) must all agree, field for field."""
    L, so = lib
    header = open(os.path.join(ROOT, "include", "emmax.h")).read()
    ctype = {"int32_t": C.c_int32, "float": C.c_float}

    def flat(fields):
        return [(n, t) for n, t in fields]

    want_tower = [(f.split("[")[0], ctype[t] * int(f.split("[")[1][:-1]) if "[" in f else ctype[t]) for t, f in _c_struct_fields(header, "emmax_tower_config")]
    assert flat(L.TowerConfigC._fields_) == want_tower
    cfg_fields = _c_struct_fields(header, "emmax_config")
    assert cfg_fields[0] == ("emmax_tower_config", "tower[2]")
    want_cfg = [("tower", L.TowerConfigC * 2)] + [(f, ctype[t]) for t, f in cfg_fields[1:]]
    assert [(n, t) for n, t in L.ConfigC._fields_][1:] == want_cfg[1:] and L.ConfigC._fields_[0][0] == "tower"
    assert so.emmax_config_size() == C.sizeof(L.ConfigC) == 2 * C.sizeof(L.TowerConfigC) + 4 * len(cfg_fields[1:])
    # the stub a maintainer would paste: execute its two struct definitions and compare
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = md[md.index("class EmmaxTower(C.Structure):"):md.index("def load(state_dict")]
    ns = {"C": C, "lib": so}   # the stub's own size / ABI assertion runs too
    exec(code, ns)
    assert [n for n, _ in ns["EmmaxTower"]._fields_] == [n for n, _ in L.TowerConfigC._fields_]
    assert [n for n, _ in ns["EmmaxConfig"]._fields_] == [n for n, _ in L.ConfigC._fields_]
    assert C.sizeof(ns["EmmaxConfig"]) == so.emmax_config_size()


def test_tuning_switches_are_a_table_with_a_setter(lib):
    """The library reads EMMAX_<NAME> once; afterwards only emmax_tuning_set moves a switch (no launcher calls getenv)."""
    L, so = lib
    names = ["graph", "ks", "ks_oproj", "ks_oproj_grid", "km", "km_down", "streamk", "fp8_gemv", "attn_nsplit", "attn_direct", "fold_embed",
             "mfma_xbar", "gemm_big", "gemm_splitk", "gemm_hybrid", "gemm_normfuse", "gemm_deep", "gemm_lnfuse", "attn_resident",
             "resid32", "kv_fp8", "km_roll", "attn_nw", "attn_deep", "attn_ksplit", "gemm_sk_big", "attn_lazy", "vis_streams", "exact"]   # (round 5: nine before the last; round 6: exact)
    header = open(os.path.join(ROOT, "include", "emmax.h")).read()
    for n in names:
        assert re.search(r"\b%s\b" % n, header), n
        L.tuning_get(n)
    assert L.tuning_get("graph") == int(os.environ.get("EMMAX_GRAPH", "0")) and L.tuning_get("ks") == int(os.environ.get("EMMAX_KS", "1"))
    assert L.tuning_get("exact") == int(os.environ.get("EMMAX_EXACT", "0"))   # exact numerics is opt-in: the headline path is the bf16-operand one
    # the product defaults of round 5: fp32 residual stream on, bf16 KV cache
    assert L.tuning_get("resid32") == int(os.environ.get("EMMAX_RESID32", "1")) and L.tuning_get("kv_fp8") == int(os.environ.get("EMMAX_KV_FP8", "0"))
    with L.tuning(graph=1, attn_nsplit=4):
        assert L.tuning_get("graph") == 1 and L.tuning_get("attn_nsplit") == 4
        os.environ["EMMAX_GRAPH"] = "0"          # the environment is not consulted again
        assert L.tuning_get("graph") == 1
        del os.environ["EMMAX_GRAPH"]
    assert L.tuning_get("graph") == 0 and L.tuning_get("attn_nsplit") == 0
    assert so.emmax_tuning_set(b"no_such_switch", 1) == -1 and b"no_such_switch" in so.emmax_last_error()
    # exactly one getenv in the library sources
    import glob

    hits = [(f, ln) for f in glob.glob(os.path.join(ROOT, "emma-x_amd", "csrc", "*.h*")) for ln in open(f) if "getenv(" in ln and not ln.lstrip().startswith("//")]
    assert len(hits) == 1 and hits[0][0].endswith("model.hip"), hits


def _model(L, so, cfg):
    from emmax.engine import _config_c

    h = C.c_void_p()
    cc = _config_c(cfg)
    return so.emmax_model_create(C.byref(cc), C.byref(h)), h


def test_config_validation_and_sizing(lib):
    from emmax.config import EmmaXConfig

    L, so = lib
    rc, h = _model(L, so, EmmaXConfig.emma_x_7b())
    assert rc == 0
    arena = so.emmax_model_arena_bytes(h)
    # main arena: all weights the path reads, bf16, once (7.53 B params minus the unused last block of each tower, plus tile
    # padding) -- what a model serving decode batches 1-2 holds (it was 37 GB whatever the batch regime)
    assert 15.0e9 < arena < 15.6e9
    # aux arena, built on demand for decode batches >= 3: qkv + gate/up in decode_km.hip's row order, o-proj / down / lm-head
    # fragment-major = one more copy of the LLM projections (6.61 B params) -> B = 8 bf16 model = 28.5 GB
    aux = so.emmax_model_aux_bytes(h)
    assert 13.0e9 < aux < 13.6e9 and arena + aux < 30e9
    # fp8 decode weights: every e4m3 copy lives in the main arena, nothing to build later
    import copy
    c8 = copy.deepcopy(EmmaXConfig.emma_x_7b())
    c8.decode_weight_dtype = "fp8"
    rc8, h8 = _model(L, so, c8)
    assert rc8 == 0 and so.emmax_model_aux_bytes(h8) == 0 and 15.0e9 < so.emmax_model_arena_bytes(h8) < 37e9
    so.emmax_model_destroy(h8)
    ws, kv = C.c_int64(), C.c_int64()
    assert so.emmax_session_bytes(h, 8, 512, 1281, C.byref(ws), C.byref(kv)) == 0
    # paged KV: 32 layers x 2 x 8 decode rows x 21 pages x 32 heads x 64 x 128 bf16 -- a plain session holds NO staging rows (ADVICE r04:
    # round 4 gave every session min(max_batch, 8) of them and doubled this region); slot serving asks for them through the _ex calls
    assert kv.value == 32 * 2 * 8 * 21 * 32 * 64 * 128 * 2
    ws2, kv2 = C.c_int64(), C.c_int64()
    assert so.emmax_session_bytes_ex(h, 8, 512, 1281, 4, C.byref(ws2), C.byref(kv2)) == 0
    assert kv2.value == 32 * 2 * (8 + 4) * 21 * 32 * 64 * 128 * 2 and ws2.value > ws.value
    assert so.emmax_session_bytes_ex(h, 8, 512, 1281, 9, C.byref(ws2), C.byref(kv2)) != 0 and b"stage_rows" in so.emmax_last_error()
    assert so.emmax_session_bytes_ex(h, 16, 512, 1281, 16, C.byref(ws2), C.byref(kv2)) == 0     # decode batches up to 16 (round 5)
    assert so.emmax_session_bytes(h, 8, 512, 700, C.byref(ws), C.byref(kv)) != 0
    assert b"max_ctx" in so.emmax_last_error()
    # exact numerics (round 6): 8 rows per projection launch (larger batches in chunks); the paged cache holds 24-bit rows (a bf16 plane + an 8-bit extension plane: 1.5 x the bf16
    # bytes), fp32 rows under exact = 2; more workspace (fp32 activations + their two-term images)
    ws1, kv1 = C.c_int64(), C.c_int64()
    assert so.emmax_session_bytes(h, 2, 512, 1281, C.byref(ws1), C.byref(kv1)) == 0
    with L.tuning(exact=1):
        assert so.emmax_session_bytes(h, 2, 512, 1281, C.byref(ws2), C.byref(kv2)) == 0
        assert kv2.value == 32 * 2 * 2 * 21 * 32 * 64 * 128 * 3 and 2 * kv2.value == 3 * kv1.value and ws2.value > ws1.value
        assert so.emmax_session_bytes(h, 8, 512, 1281, C.byref(ws2), C.byref(kv2)) == 0 and kv2.value == 32 * 2 * 8 * 21 * 32 * 64 * 128 * 3
        assert so.emmax_session_bytes(h, 16, 512, 1281, C.byref(ws2), C.byref(kv2)) == 0     # above 8 rows: the projections run in chunks of 8
        assert so.emmax_session_bytes_ex(h, 4, 512, 1281, 2, C.byref(ws2), C.byref(kv2)) == 0 and kv2.value == 32 * 2 * (4 + 2) * 21 * 32 * 64 * 128 * 3   # staging rows: slot serving
    with L.tuning(exact=2):
        assert so.emmax_session_bytes(h, 2, 512, 1281, C.byref(ws2), C.byref(kv2)) == 0 and kv2.value == 2 * kv1.value
    assert L.tuning_get("exact") == 0
    so.emmax_model_destroy(h)
    bad = EmmaXConfig.tiny()
    bad.llm.head_dim = 64
    rc, _ = _model(L, so, bad)
    assert rc == -1 and b"head_dim" in so.emmax_last_error()
    bad = EmmaXConfig.tiny()
    bad.towers[1].embed_dim = 160   # head_dim 80
    rc, _ = _model(L, so, bad)
    assert rc == -1 and b"tower 1" in so.emmax_last_error()


def test_null_arguments_are_errors(lib):
    L, so = lib
    assert so.emmax_model_create(None, None) == -1
    assert so.emmax_decode_step(None, None) == -1
    assert so.emmax_generate(None, 4, 1, None, None, None) == -1


def test_gemm_launch_plans_of_the_hot_path(lib):
    """emmax_gemm_plan (host only): the launch plans of the GEMM shapes of the hot path, pinned -- one-frame prefill (M = 768: gate/up =
    one round of 256x256 tiles + a K-split column remainder, qkv one under-filled round, o / down K-split with the RMSNorm in the reduce
    pass), eight frames (M = 6144), ViT at 256 frames (row plan, the half-empty SigLIP tile column), and what the switches turn off."""
    L, so = lib
    assert L.gemm_plan(768, 22016, 4096, act=2) == "hybrid cols 0..21760: big | cols 21760..22016: splitk ks=8"
    assert L.gemm_plan(768, 22016, 4096, act=2, ws_bytes=0) == "big rows 0..512 + small rows 512..768"      # no scratch: never split
    assert L.gemm_plan(768, 12288, 4096) == "big"
    assert L.gemm_plan(768, 4096, 4096, residual=True, norm=True) == "splitk ks=2 +norm"
    # (round 5: a VERY long K and at most half a round of 256x256 tiles -- one slice of a big tile per CU; tools/gemm_sk_sweep.py)
    assert L.gemm_plan(768, 4096, 11008, residual=True, norm=True) == "splitk big ks=5 +norm"
    assert L.gemm_plan(768, 4096, 11008, residual=True) == "splitk big ks=5"
    assert L.gemm_plan(1536, 4096, 11008, residual=True, norm=True) == "splitk big ks=2 +norm"      # two frames: 96 big tiles
    assert L.gemm_plan(1536, 4096, 4096, residual=True, norm=True) == L.gemm_plan(1536, 4096, 4096, residual=True)   # K = 4096: no split from 384 small tiles on
    assert L.gemm_plan(512, 4096, 8704, act=1) == "splitk ks=4"                                       # the projector's fc2 at two frames: 32 big tiles, small ones win
    with L.tuning(gemm_sk_big=0):
        assert L.gemm_plan(768, 4096, 11008, residual=True, norm=True) == "splitk ks=2 +norm"
    # round 5: the prefill's fp32 residual stream (fp32 residual in, fp32 C out) takes the same plans, the norm still in the reduce pass
    assert L.gemm_plan(768, 4096, 4096, out_f32=True, residual=2, norm=True) == "splitk ks=2 +norm"
    assert L.gemm_plan(768, 4096, 11008, out_f32=True, residual=2, norm=True) == "splitk big ks=5 +norm"
    assert L.gemm_plan(6144, 4096, 11008, out_f32=True, residual=2, norm=True) == "big rows 0..4096 + small rows 4096..6144"
    assert L.gemm_plan(768, 2048, 4096, residual=True, norm=True) == "splitk ks=5"                             # the fused norm is for 4096-wide rows
    assert L.gemm_plan(6144, 22016, 4096, act=2) == "hybrid cols 0..21760: big | cols 21760..22016: splitk ks=5"
    assert L.gemm_plan(6144, 4096, 4096, residual=True, norm=True) == "big rows 0..4096 + small rows 4096..6144"
    assert L.gemm_plan(66816, 1024, 4096, residual=True) == "big rows 0..62720 + small rows 62720..66816"
    assert L.gemm_plan(66816, 4096, 1024, act=1, ln=True) == "big rows 0..64768 + small rows 64768..66816"
    assert L.gemm_plan(65536, 1152, 1152, residual=True) == "cols 0..1024: big | cols 1024..1152: small"
    assert L.gemm_plan(65536, 1152, 4352, residual=True) == "big"                                              # the narrow launch would re-read all of A
    assert L.gemm_plan(261, 1024, 4096, residual=True) == "splitk ks=8"                                        # batch-1 ViT fc2
    assert L.gemm_plan(8192, 8192, 8192) == "big"
    with L.tuning(gemm_hybrid=0):
        assert L.gemm_plan(768, 22016, 4096, act=2) == "big rows 0..512 + small rows 512..768"
    with L.tuning(gemm_normfuse=0):
        assert L.gemm_plan(768, 4096, 4096, residual=True, norm=True) == "splitk ks=2"
    with L.tuning(gemm_splitk=0):
        assert L.gemm_plan(768, 4096, 4096, residual=True, norm=True) == "small"
        assert L.gemm_plan(768, 22016, 4096, act=2) == "big rows 0..512 + small rows 512..768"
    with L.tuning(gemm_big=1):
        assert L.gemm_plan(768, 4096, 4096) == "forced big"
    assert so.emmax_gemm_plan(768, 100, 4096, 0, 0, 0, 0, 0, 0, None, 0) != 0 and so.emmax_gemm_plan(768, 100, 4096, 0, 0, 0, 0, 0, 0, b" " * 64, 64) != 0
