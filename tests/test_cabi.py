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
    assert so.emmax_abi_version() == 1 and b"gfx950" in so.emmax_version()


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
    # all weights the path reads, bf16 (7.53 B params minus the unused last block of each tower, plus tile padding)
    # + the MFMA-fragment-major copy of the LLM projections used by the batch >= 3 decode path (6.74 B params)
    # + the row-permuted fragment-major copy of qkv and gate/up for the K-split MFMA kernel (decode_km.hip: 4.5 B params)
    assert 36.8e9 < arena < 37.8e9
    ws, kv = C.c_int64(), C.c_int64()
    assert so.emmax_session_bytes(h, 8, 512, 1281, C.byref(ws), C.byref(kv)) == 0
    # paged KV: 32 layers x 2 x 8 rows x 21 pages x 32 heads x 64 x 128 bf16
    assert kv.value == 32 * 2 * 8 * 21 * 32 * 64 * 128 * 2
    assert so.emmax_session_bytes(h, 8, 512, 700, C.byref(ws), C.byref(kv)) != 0
    assert b"max_ctx" in so.emmax_last_error()
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
