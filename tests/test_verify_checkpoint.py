"""tests/verify_checkpoint.py (the one-command real-checkpoint check) on the synthetic HF-format checkpoint:
CPU: file / state-dict validation and the tokenizer section (driven with the stub tokenizer); GPU: the verify_openvla-style
model section -- HIP `predict_action` vs the oracle over 3 random 256x256 images."""

import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    from emmax.config import EmmaXConfig
    from tools.make_synthetic_checkpoint import write_checkpoint

    d = str(tmp_path_factory.mktemp("ckpt"))
    write_checkpoint(d, EmmaXConfig.tiny(), seed=2, planted=True, tiny_towers=True)
    return d


def test_files_and_tokenizer_sections(ckpt, tmp_path, monkeypatch):
    import verify_checkpoint as vc

    out = str(tmp_path / "r.json")
    assert vc.main([ckpt, "--skip-model", "--json", out]) == 0
    rep = json.load(open(out))
    assert rep["files"]["status"] == "PASS" and rep["tokenizer"]["status"] == "SKIPPED" and rep["model"]["status"] == "SKIPPED"
    assert rep["timm"]["status"] in ("SKIPPED", "PASS")
    # the tokenizer section itself, with the stub standing in for the LLaMA files
    import emmax.modeling as modeling
    from emmax.tokenizer_stub import StubTokenizer

    monkeypatch.setattr(modeling, "load_tokenizer", lambda path, cfg: StubTokenizer())
    assert vc.main([ckpt, "--skip-model", "--json", out]) == 0
    tok = json.load(open(out))["tokenizer"]
    assert tok["status"] == "PASS" and tok["round_trip_worst"] < 1e-2 and 29871 not in tok["stop_trigger"]
    # a broken directory fails loudly
    os.remove(os.path.join(ckpt, "model-00001-of-00002.safetensors"))
    assert vc.main([ckpt, "--skip-model", "--json", out]) == 1
    assert json.load(open(out))["files"]["status"] == "FAIL"


@pytest.mark.gpu
def test_model_section_on_the_hip_path(device, tmp_path):
    import verify_checkpoint as vc
    from emmax.config import EmmaXConfig
    from tools.make_synthetic_checkpoint import write_checkpoint

    d = str(tmp_path / "ckpt")
    write_checkpoint(d, EmmaXConfig.tiny(), seed=2, planted=True, tiny_towers=True)
    out = str(tmp_path / "r.json")
    assert vc.main([d, "--device", device, "--prompts", "3", "--json", out]) == 0
    rep = json.load(open(out))["model"]
    assert rep["status"] == "PASS" and len(rep["prompts"]) == 3
    assert all(r["ids_equal"] and r["action_err"] <= 1e-3 for r in rep["prompts"])
    assert all(r["logit_rel_err_max"] < 3e-2 and r["teacher_forced_flips"] == 0 for r in rep["prompts"])   # planted margins: no flip in either mode
    # the same command with the HIP side in exact numerics (round 6): the margin statistic drops to fp32 level
    assert vc.main([d, "--device", device, "--prompts", "2", "--exact", "--json", out]) == 0
    rep = json.load(open(out))["model"]
    assert rep["status"] == "PASS" and rep["hip_numerics"].startswith("exact")
    assert all(r["ids_equal"] and r["logit_rel_err_max"] < 1e-4 and r["teacher_forced_flips"] == 0 for r in rep["prompts"])
