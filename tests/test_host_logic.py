"""CPU: host-side logic of the product package (no device work): config/weight contracts, processor, planted chain,
readers, error behaviour, and that the product path refuses to run without the HIP device."""

import json
import math
import os

import numpy as np
import pytest
import torch

from emmax.config import EmmaXConfig


def test_7b_shapes_and_param_count():
    from emmax.weights import param_shapes

    cfg = EmmaXConfig.emma_x_7b()
    shapes = {k: s for k, s, _ in param_shapes(cfg)}
    assert shapes["language_model.lm_head.weight"] == (32064, 4096)
    assert shapes["vision_backbone.featurizer.reg_token"] == (1, 4, 1024)
    assert shapes["vision_backbone.fused_featurizer.blocks.26.mlp.fc1.weight"] == (4304, 1152)
    assert shapes["vision_backbone.fused_featurizer.blocks.0.attn.qkv.weight"] == (3456, 1152)
    assert shapes["projector.fc1.weight"] == (8704, 2176)
    assert "vision_backbone.fused_featurizer.blocks.0.ls1.scale_factor" not in shapes
    total = sum(math.prod(s) for s in shapes.values())
    assert abs(total - 7.526e9) < 2e7
    assert cfg.towers[0].take_index == 22 and cfg.towers[1].take_index == 25
    assert cfg.towers[0].n_tokens == 261 and cfg.towers[1].n_tokens == 256 and cfg.towers[1].head_dim == 72
    assert cfg.action_vocab_size == 32000 and cfg.projector_dims == (2176, 8704, 4096, 4096)


def test_config_from_hf_dict_and_rejections(tmp_path):
    d = {"vision_backbone_id": "dinosiglip-vit-so-224px", "llm_backbone_id": "llama2-7b-pure",
         "text_config": {"hidden_size": 4096, "num_hidden_layers": 32, "num_attention_heads": 32, "rms_norm_eps": 1e-5,
                         "vocab_size": 32064, "intermediate_size": 11008}, "n_action_bins": 256,
         "norm_stats": {"bridge_orig": {"action": {"q01": [0] * 7, "q99": [1] * 7}}}}
    cfg = EmmaXConfig.from_hf_dict(d)
    assert cfg.llm.num_kv_heads == 32 and cfg.llm.head_dim == 128 and cfg.llm.rms_eps == 1e-5
    with pytest.raises(ValueError):
        EmmaXConfig.from_hf_dict({**d, "vision_backbone_id": "clip-vit-l-336px"})
    with pytest.raises(ValueError):
        EmmaXConfig.from_hf_dict({**d, "llm_backbone_id": "mistral-v0.1-7b-pure"})
    (tmp_path / "config.json").write_text(json.dumps(d))
    (tmp_path / "dataset_statistics.json").write_text(json.dumps({"x": {"action": {"q01": [0] * 7, "q99": [2] * 7}}}))
    assert list(EmmaXConfig.from_pretrained(str(tmp_path)).norm_stats) == ["x"]


def test_synthetic_weights_deterministic_and_planted_chain():
    from emmax.weights import planted_chain, planted_start_token, planted_successor, synthetic_state_dict

    cfg = EmmaXConfig.tiny()
    a, b = synthetic_state_dict(cfg, seed=3), synthetic_state_dict(cfg, seed=3)
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = synthetic_state_dict(cfg, seed=4)
    assert not torch.equal(a["projector.fc1.weight"], c["projector.fc1.weight"])
    succ = planted_successor(cfg)
    start = planted_start_token(cfg, 5)
    chain = planted_chain(cfg, start, 64)
    assert chain[4] == 29871 and chain[-1] == cfg.eos_token_id and len(chain) == 5 + 8 + 1
    assert all(31744 <= t < 32000 for t in chain[5:13]) and len(set(chain[5:13])) == 8
    assert int(succ[1]) == 29871


def test_safetensors_and_native_readers(tmp_path):
    from safetensors.torch import save_file

    from emmax.weights import (load_hf_state_dict, remap_native_state_dict, synthetic_state_dict, validate_state_dict)

    cfg = EmmaXConfig.tiny()
    sd = {k: v.to(torch.bfloat16) for k, v in synthetic_state_dict(cfg, seed=1).items()}
    keys = sorted(sd)
    save_file({k: sd[k].contiguous() for k in keys[: len(keys) // 2]}, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file({k: sd[k].contiguous() for k in keys[len(keys) // 2:]}, str(tmp_path / "model-00002-of-00002.safetensors"))
    back = load_hf_state_dict(str(tmp_path))
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    validate_state_dict(back, cfg)
    bad = dict(back)
    bad.pop("projector.fc2.bias")
    bad["projector.fc1.bias"] = torch.zeros(3)
    with pytest.raises(ValueError) as e:
        validate_state_dict(bad, cfg)
    assert "missing projector.fc2.bias" in str(e.value) and "projector.fc1.bias" in str(e.value)
    # native .pt layout -> HF keys (convert_openvla_weights_to_hf.py:84-116)
    native = {"projector": {}, "llm_backbone": {}, "vision_backbone": {}}
    for k, v in sd.items():
        if k.startswith("projector."):
            idx = {"fc1": "0", "fc2": "2", "fc3": "4"}[k.split(".")[1]]
            native["projector"][f"projector.{idx}.{k.split('.')[2]}"] = v
        elif k.startswith("language_model."):
            native["llm_backbone"]["llm." + k[len("language_model."):]] = v
        elif k.startswith("vision_backbone.featurizer."):
            native["vision_backbone"]["dino_featurizer." + k[len("vision_backbone.featurizer."):].replace("scale_factor", "gamma")] = v
        else:
            native["vision_backbone"]["siglip_featurizer." + k[len("vision_backbone.fused_featurizer."):]] = v
    again = remap_native_state_dict(native)
    assert set(again) == set(sd) and all(torch.equal(again[k], sd[k]) for k in sd)


def test_processor_matches_oracle_preprocessing():
    from emmax.processing import EmmaXProcessor
    from oracle import emmax_oracle as orc

    cfg = EmmaXConfig.tiny()
    proc = EmmaXProcessor.from_pretrained(cfg=cfg)
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, size=(2, 224, 224, 3), dtype=np.uint8)
    out = proc(["hello", "hellp"], [frames[0], frames[1]])
    assert out["pixel_values"].shape == (2, 6, 224, 224) and out["pixel_values"].dtype == torch.float32
    assert torch.equal(out["pixel_values"], orc.preprocess_frames(frames, cfg))
    assert torch.equal(out["frames_u8"], torch.from_numpy(frames))
    assert out["input_ids"][0, 0] == 1 and out["input_ids"].shape == out["attention_mask"].shape
    moved = out.to("cpu", dtype=torch.bfloat16)
    assert moved["pixel_values"].dtype == torch.bfloat16 and moved["input_ids"].dtype == torch.long
    with pytest.raises(ValueError):
        proc(["only one"], [frames[0], frames[1]])
    prompt, img = proc.get_prompt("put the carrot on the plate", frames[0])
    assert prompt == "In: What action should the robot take to achieve the instruction\nINSTRUCTION: \nput the carrot on the plate\nOut:"
    # non-native resolution goes through PIL bicubic (torchvision's TVF.resize on PIL == PIL resize)
    big = rng.integers(0, 256, size=(256, 256, 3), dtype=np.uint8)
    assert proc.image_processor.preprocess(big)["pixel_values"].shape == (1, 6, 224, 224)


def test_stub_tokenizer_roundtrip():
    from emmax.tokenizer_stub import StubTokenizer

    tok = StubTokenizer()
    ids = tok("POLICIES:\nab", add_special_tokens=True).input_ids
    assert ids[0] == 1 and ids[1] == 29871
    assert tok.decode(ids, skip_special_tokens=True) == "POLICIES:\nab"
    act = list(range(31744, 32000))
    assert tok(tok.decode(act), add_special_tokens=False).input_ids[1:] == act
    assert tok.decode([2, 32000, 5], skip_special_tokens=True) == tok.decode([5])


def test_product_refuses_cpu_and_missing_extension(monkeypatch):
    from emmax import _lib
    from emmax.modeling import EmmaXForActionPrediction
    from emmax.weights import synthetic_state_dict

    cfg = EmmaXConfig.tiny()
    m = EmmaXForActionPrediction(cfg, synthetic_state_dict(cfg, 0))
    with pytest.raises(RuntimeError, match="no CPU"):
        m.to("cpu")
    with pytest.raises(RuntimeError, match="not on a HIP device"):
        m.generate(torch.tensor([[1, 5]]), pixel_values=torch.zeros(1, 6, 224, 224))
    with pytest.raises(NotImplementedError):
        EmmaXForActionPrediction.from_pretrained("/nonexistent", load_in_8bit=True)
    with pytest.raises(ValueError, match="unnorm_key"):
        m._check_unnorm_key({"a": {}, "b": {}}, None)
    assert m.get_action_dim("bridge_orig") == 7
    # a missing shared library is a hard error, never a fallback
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libemmax_hip.so")
    with pytest.raises(_lib.EmmaxError, match="no CPU fallback"):
        _lib.load()


def test_ids_level_action_extraction():
    from emmax.modeling import EmmaXForActionPrediction
    from oracle import emmax_oracle as orc

    cfg = EmmaXConfig.tiny()
    m = EmmaXForActionPrediction(cfg, None)
    stats = cfg.norm_stats["bridge_orig"]["action"]
    mv = list(range(31800, 31808))
    pol = [31900, 31800, 31850, 31760, 31990, 31872, 31871, 31745]
    row = [50, 60] + mv + [13, 29871] + pol + [2]
    got = m.actions_from_ids(row, stats)
    ref = orc.unnormalize_actions(orc.decode_token_ids_to_actions(np.array(pol[:7])), stats)
    assert np.abs(got - ref).max() < 1e-6
    assert np.array_equal(m.actions_from_ids([5, 6, 7], stats), np.zeros(7, dtype=np.float32))


def test_bicubic_tables_match_pillow():
    """The coefficient restatement (emmax/resize.py) against the installed Pillow, incl. the robot's 256 -> 224."""
    from PIL import Image

    from emmax.resize import bicubic_coeffs, resize_u8_reference

    rng = np.random.default_rng(0)
    for h, w in [(256, 256), (480, 640), (224, 300), (100, 180)]:
        a = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        assert np.array_equal(resize_u8_reference(a, 224, 224), np.asarray(Image.fromarray(a).resize((224, 224), Image.BICUBIC)))
    b, k, n = bicubic_coeffs(256, 224)
    assert n == 7 and b.shape == (224, 2) and abs(int(k[100].sum()) - (1 << 22)) <= 4


def test_real_tokenizer_files_are_picked_up_and_give_a_stop_rule(tmp_path):
    """A checkpoint directory with tokenizer files: the processor loads them through `transformers.AutoTokenizer` (a tiny
    word-level vocabulary here: the LLaMA files are not available offline, so the real text<->id parity stays unpinned) and
    `stop_rule_from_tokenizer` turns the POLICIES: marker into the device-side stop rule."""
    tokenizers = pytest.importorskip("tokenizers")
    pytest.importorskip("transformers")
    import json

    from emmax.config import EmmaXConfig
    from emmax.processing import EmmaXProcessor
    from emmax.serving import stop_rule_from_tokenizer

    words = ["<unk>", "<s>", "</s>", "In:", "Out:", "What", "action", "should", "the", "robot", "take", "to", "pick", "up", "cup", "?",
             "POLICIES:", "MOVE", "GRIPPER", "\n"]
    vocab = {w: i for i, w in enumerate(words)}
    tk = tokenizers.Tokenizer(tokenizers.models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = tokenizers.pre_tokenizers.WhitespaceSplit()
    tk.save(str(tmp_path / "tokenizer.json"))
    (tmp_path / "tokenizer_config.json").write_text(json.dumps({
        "tokenizer_class": "PreTrainedTokenizerFast", "bos_token": "<s>", "eos_token": "</s>", "unk_token": "<unk>", "pad_token": "</s>"}))
    cfg = EmmaXConfig.tiny()
    proc = EmmaXProcessor.from_pretrained(str(tmp_path), cfg=cfg)
    assert type(proc.tokenizer).__name__ != "StubTokenizer"
    img = np.zeros((224, 224, 3), dtype=np.uint8)
    feat = proc("What action should the robot take to pick up the cup ?", img)
    ids = feat["input_ids"][0].tolist()
    assert ids == [vocab[w] for w in "What action should the robot take to pick up the cup ?".split()]
    assert proc.decode(ids) == "What action should the robot take to pick up the cup ?"
    trig, after = stop_rule_from_tokenizer(proc.tokenizer)
    assert trig == [vocab["POLICIES:"]] and after == 8
    with pytest.raises(ValueError):
        stop_rule_from_tokenizer(proc.tokenizer, marker=" ".join(["cup"] * 17))


def test_bench_contract_flags_and_committed_bench_line():
    """The driver's contract for bench.py: the flags it passes exist, and the bench line committed under profiles/ carries every
    required key (incl. the `roofline` and `cpu_baseline` objects) with sane values."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
    line = open(os.path.join(ROOT, "profiles", "r02_bench_n1.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "actions/sec" and d["unit"] == "actions/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-3 * d["value"]          # one action per step at N = 1, B = 1
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.4 < r["frac"] < 1.0
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["bytes_per_launch"] < 1.2
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "actions/s" and c["sample"]
    assert "executed in full" in c["sample"] and c["parts_s"]["prefill_layers"] > 0 and c["parts_s"]["decode_step_mean"] > 0
    assert d["rccl_ranks"] == 1 and d["gather_ms"] >= 0 and d["config"]["workload"].startswith("BASELINE configs[1]")


def test_bench_gpus_n_without_a_launcher_re_executes_under_torchrun():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must become the driver's own multi-process launch line instead
    of dying on the WORLD_SIZE check.  Without a GPU every rank then stops at the device check -- once per rank."""
    import subprocess
    import sys

    from conftest import ROOT

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--tiny", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0
    assert "WORLD_SIZE=" not in out.stderr
    assert out.stderr.count("bench.py needs a HIP device") == 2


def test_stop_rule_is_tokenised_in_context_sentencepiece_style():
    """a SentencePiece / LLaMA tokenizer prepends a dummy `▁` to a bare string, giving `▁POLICIES` -- a piece that
    never follows `\\n` in generated text.  The stop rule must be what the model emits after a newline."""
    tokenizers = pytest.importorskip("tokenizers")
    from tokenizers import normalizers

    from emmax.serving import stop_rule_from_tokenizer
    from emmax.tokenizer_stub import StubTokenizer

    pieces = ["<unk>", "▁", "\n", "POLICIES", "▁POLICIES", ":", "MOVEMENT", "▁MOVEMENT"] + [chr(c) for c in range(65, 91)]
    tk = tokenizers.Tokenizer(tokenizers.models.Unigram([(p, -1.0 if len(p) > 1 else -5.0) for p in pieces], unk_id=0))
    tk.normalizer = normalizers.Sequence([normalizers.Prepend("▁"), normalizers.Replace(" ", "▁")])

    class Wrap:   # the HF call surface stop_rule_from_tokenizer uses
        def __call__(self, text, add_special_tokens=False):
            return {"input_ids": tk.encode(text, add_special_tokens=add_special_tokens).ids}

    vocab = {p: i for i, p in enumerate(pieces)}
    bare = Wrap()("POLICIES:")["input_ids"]
    assert bare == [vocab["▁POLICIES"], vocab[":"]]                  # what the old rule armed: never emitted after "\n"
    trig, after = stop_rule_from_tokenizer(Wrap())
    assert trig == [vocab["POLICIES"], vocab[":"]] and after == 8
    trig, _ = stop_rule_from_tokenizer(StubTokenizer())              # stub: no leading 29871 either
    assert 29871 not in trig and StubTokenizer().decode(trig) == "POLICIES:"


def test_norm_stats_are_never_invented(tmp_path):
    """a checkpoint without statistics must not un-normalise with made-up values."""
    from emmax.modeling import EmmaXForActionPrediction

    assert EmmaXConfig.emma_x_7b().norm_stats == {}
    d = {"vision_backbone_id": "dinosiglip-vit-so-224px", "llm_backbone_id": "llama2-7b-pure", "text_config": {}}
    cfg = EmmaXConfig.from_hf_dict(d)
    assert cfg.norm_stats == {}
    m = EmmaXForActionPrediction(cfg, None)
    with pytest.raises(ValueError):
        m.get_action_stats(None)
    with pytest.raises(ValueError):
        m.get_action_dim("bridge_orig")
    assert "bridge_orig" in EmmaXConfig.tiny().norm_stats            # synthetic factories keep their synthetic statistics
    # native `.pt`: the run directory must hold config.json AND dataset_statistics.json (prismatic/models/load.py:133-144)
    run = tmp_path / "run" / "checkpoints"
    run.mkdir(parents=True)
    pt = run / "step-000001.pt"
    torch.save({"model": {}}, str(pt))
    with pytest.raises(FileNotFoundError, match="config.json"):
        EmmaXForActionPrediction.from_pretrained(str(pt))
    (tmp_path / "run" / "config.json").write_text(json.dumps({"vla": {"base_vlm": "prism-dinosiglip-224px+7b"}}))
    with pytest.raises(FileNotFoundError, match="dataset_statistics.json"):
        EmmaXForActionPrediction.from_pretrained(str(pt))
    bad = tmp_path / "elsewhere.pt"
    torch.save({"model": {}}, str(bad))
    with pytest.raises(ValueError, match="Invalid checkpoint"):
        EmmaXForActionPrediction.from_pretrained(str(bad))


def test_generate_actions_pos_tail_on_a_textual_movement_line():
    """VERDICT r04 missing #5: `generate_actions(type="pos")` when the MOVEMENT line is textual ("move forward 3; ...": solver.py:57-58,
    require_unorm False) or unparsable (require_unorm None).  The reference's tail (prismatic.py:686-696) only assigns `proprio_norm`
    under `if require_unorm:` and dies with UnboundLocalError otherwise -- pinned here on the oracle's branch-for-branch restatement.
    The product's documented repair: the Solver's delta is returned unchanged -- it is already in
    physical units on the textual branch, and the `[-100] * 7` sentinel on the failure branch.  On the tokenised branch both agree."""
    from emmax.modeling import EmmaXForActionPrediction
    from emmax.tokenizer_stub import StubTokenizer
    from oracle import emmax_oracle as orc

    from conftest import GOLDEN

    cases = json.load(open(os.path.join(GOLDEN, "solver.json"), encoding="utf-8"))
    m = EmmaXForActionPrediction(EmmaXConfig.tiny(), None)
    tok = StubTokenizer()
    stats = m.get_proprio_stats()
    seen = set()
    for c in cases:
        req, mv = c["require_unorm"], c["movement"]       # what the REFERENCE Solver returned for this text (golden G7)
        seen.add(req)
        got, _ = m._postprocess(tok(c["text"], add_special_tokens=False).input_ids, tok, type="pos")
        if req:
            np.testing.assert_allclose(np.asarray(got, dtype=np.float64), orc.generate_actions_pos_tail(req, mv, stats), rtol=0, atol=1e-12)
        else:
            with pytest.raises(UnboundLocalError):       # the reference's behaviour on this branch
                orc.generate_actions_pos_tail(req, mv, stats)
            assert list(got) == list(mv)                  # the product: the Solver's delta, unchanged
    assert seen == {True, False, None}, seen


def test_generate_length_arguments_follow_hf():
    """`max_length` is a TOTAL (prompt included) and yields to `max_new_tokens`; `min_length` beyond the prompt would need EOS
    suppression and raises."""
    from emmax.modeling import EmmaXForActionPrediction

    m = EmmaXForActionPrediction(EmmaXConfig.tiny(), None)
    rows = [[1] * 30, [1] * 12]
    assert m._max_new(rows, 512, None, 1) == 512
    assert m._max_new(rows, None, 100, None) == 70
    assert m._max_new(rows, 7, 100, None) == 7
    assert m._max_new(rows, None, 10, None) == 1
    assert m._max_new(rows, None, None, None) == 20
    with pytest.raises(NotImplementedError):
        m._max_new(rows, 64, None, 40)


def test_register_auto_classes_resolves_to_the_mi355x_classes(tmp_path):
    """The reference's plug-in point (experiments/robot/openvla_utils.py:38-41): after `emmax.register_auto_classes()` the
    Auto classes resolve an `openvla` checkpoint directory to the MI355X model / processor (no device work here)."""
    transformers = pytest.importorskip("transformers")
    import sys

    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import emmax
    from emmax.modeling import EmmaXForActionPrediction
    from emmax.processing import EmmaXProcessor
    from tools.make_synthetic_checkpoint import write_checkpoint

    report = emmax.register_auto_classes()
    assert report["AutoConfig"] == "ok" and report["AutoProcessor"] == "ok"
    assert emmax.register_auto_classes()["AutoConfig"] == "ok"       # idempotent
    ck = str(tmp_path / "ckpt")
    write_checkpoint(ck, EmmaXConfig.tiny(), seed=1, tiny_towers=True)
    cfg = transformers.AutoConfig.from_pretrained(ck)
    assert cfg.model_type == "openvla" and type(cfg).__name__ == "OpenVLAConfig"
    auto = getattr(transformers, "AutoModelForVision2Seq", None) or transformers.AutoModelForImageTextToText
    assert report[auto.__name__] == "ok"
    kw = {"dtype": torch.bfloat16} if int(transformers.__version__.split(".")[0]) >= 5 else {"torch_dtype": torch.bfloat16}
    vla = auto.from_pretrained(ck, low_cpu_mem_usage=True, trust_remote_code=True, **kw)
    assert isinstance(vla, EmmaXForActionPrediction) and list(vla.norm_stats) == ["bridge_orig"]
    with pytest.raises(FileNotFoundError):            # no tokenizer files in the directory: no silent stand-in
        transformers.AutoProcessor.from_pretrained(ck, trust_remote_code=True)
    _write_wordlevel_tokenizer(ck)
    proc = transformers.AutoProcessor.from_pretrained(ck, trust_remote_code=True)
    assert isinstance(proc, EmmaXProcessor) and type(proc.tokenizer).__name__ != "StubTokenizer"
    with pytest.raises(RuntimeError, match="MI355X"):
        vla.to("cpu")


def _write_wordlevel_tokenizer(d):
    tokenizers = pytest.importorskip("tokenizers")
    import json
    words = ["<unk>", "<s>", "</s>", "In:", "Out:", "What", "action", "should", "the", "robot", "take", "to", "pick", "up", "cup", "?"]
    tk = tokenizers.Tokenizer(tokenizers.models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
    tk.pre_tokenizer = tokenizers.pre_tokenizers.WhitespaceSplit()
    tk.save(os.path.join(d, "tokenizer.json"))
    with open(os.path.join(d, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "bos_token": "<s>", "eos_token": "</s>", "unk_token": "<unk>",
                   "pad_token": "</s>"}, f)


def test_checkpoint_with_auto_map_still_resolves_to_the_mi355x_class(tmp_path):
    """Real checkpoints carry `auto_map` + bundled modeling code; with trust_remote_code=True transformers 4.x would load THAT.
    `load_vision2seq` forces the local class and checks the resolved type; `require_emmax` is the check on its own."""
    pytest.importorskip("transformers")
    import sys

    from conftest import ROOT
    sys.path.insert(0, ROOT)
    from emmax.hf_auto import load_vision2seq, require_emmax
    from emmax.modeling import EmmaXForActionPrediction
    from tools.make_synthetic_checkpoint import write_checkpoint

    ck = str(tmp_path / "ckpt")
    write_checkpoint(ck, EmmaXConfig.tiny(), seed=2, tiny_towers=True, auto_map=True)
    import json
    assert "auto_map" in json.load(open(os.path.join(ck, "config.json")))
    vla = load_vision2seq(ck, torch_dtype=torch.bfloat16, low_cpu_mem_usage=True, trust_remote_code=True)
    assert isinstance(vla, EmmaXForActionPrediction)
    with pytest.raises(RuntimeError, match="auto_map"):
        require_emmax(torch.nn.Linear(2, 2))


def test_full_size_synthetic_checkpoint_config_carries_statistics(tmp_path, monkeypatch):
    """tools/make_synthetic_checkpoint.py --full used to write empty norm_stats (config default) -> predict_action raised."""
    import sys

    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import tools.make_synthetic_checkpoint as mk

    monkeypatch.setattr(mk, "synthetic_state_dict", lambda cfg, seed=0, planted=True: {"w": torch.zeros(2, 2)})
    cfg = EmmaXConfig.emma_x_7b()
    assert not cfg.norm_stats
    mk.write_checkpoint(str(tmp_path), cfg, shards=1)
    import json
    assert list(json.load(open(tmp_path / "dataset_statistics.json"))) == ["bridge_orig"]
    assert list(json.load(open(tmp_path / "config.json"))["norm_stats"]) == ["bridge_orig"]


def test_generate_actions_accepts_a_userdict_batchfeature_and_processor_needs_a_tokenizer(tmp_path):
    """transformers.BatchFeature is a UserDict, not a dict (processing_prismatic.py:216): the README form must still be taken.
    No device here: the route is observed through the first thing each form touches."""
    from collections import UserDict

    from emmax.modeling import EmmaXForActionPrediction
    from emmax.processing import EmmaXImageProcessor, EmmaXProcessor

    m = EmmaXForActionPrediction(EmmaXConfig.tiny(), None)
    feat = UserDict(input_ids=torch.tensor([[1, 5, 6]]), attention_mask=torch.ones(1, 3, dtype=torch.long), pixel_values=torch.zeros(1, 6, 224, 224))
    with pytest.raises(ValueError, match="no tokenizer given"):     # README branch reached (native form would raise TypeError)
        m.generate_actions(feat, do_sample=False, max_new_tokens=4)
    with pytest.raises(TypeError):
        m.generate_actions(torch.zeros(3), do_sample=False)          # not a mapping with input_ids: native form, arguments missing
    with pytest.raises(ValueError, match="needs a tokenizer"):
        EmmaXProcessor(EmmaXImageProcessor(EmmaXConfig.tiny()), None)
    with pytest.raises(FileNotFoundError, match="tokenizer"):
        (tmp_path / "config.json").write_text("{}")
        EmmaXProcessor.from_pretrained(str(tmp_path), cfg=EmmaXConfig.tiny())
    assert type(EmmaXProcessor.from_synthetic(EmmaXConfig.tiny()).tokenizer).__name__ == "StubTokenizer"
    assert type(EmmaXProcessor.from_pretrained(cfg=EmmaXConfig.tiny()).tokenizer).__name__ == "StubTokenizer"   # no path = synthetic
