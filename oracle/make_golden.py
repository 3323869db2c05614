"""
make_golden.py -- generate the golden vectors under tests/golden/ from the REAL reference pieces.

Runs ONLY in the authoring container (it reads /root/reference and imports `transformers`); the GPU box never runs it.
It commits DATA (inputs + expected outputs), never reference source.  What it pins:

  G1 action_decode.npz      ids -> actions through the reference's ActionTokenizer (prismatic/vla/action_tokenizer.py:49-68)
  G2 (inside G5)            un-normalisation through OpenVLAForActionPrediction.predict_action (modeling_prismatic.py:522-535)
  G3 projector.npz          reference FusedMLPProjector (prismatic/util/nn_utils.py:37-53), seeded weights stored
  G4 llama_{mha,gqa}.npz    HF LlamaForCausalLM CPU fp32: prefill logits, per-step decode logits, greedy ids (weights stored)
  G5 wrapper.npz            reference OpenVLAForActionPrediction (modeling_prismatic.py) run through a stub-timm shim:
                            forward logits, predict_action ids + 7-vector  (weights = emmax synthetic seed, checksum stored)
  G6 prompts.json           PurePromptBuilder strings (base_prompter.py:28-73)
  G7 solver.json            Solver.extract_action_policies / extract_movement_plan (solver.py:8-137) with a stub tokenizer
  G8 vit_hf.npz             the two ViT towers against an INDEPENDENT implementation of the same published architectures:
                            transformers' Dinov2WithRegistersModel (= timm vit_*_patch14_reg4_dinov2) and SiglipVisionModel
                            (= timm vit_so400m_patch14_siglip) with exact-erf GELU, fed the emmax synthetic weights (seed
                            stored); expected = hidden state after block `take_index`, prefix tokens dropped.  timm itself
                            (the reference's dependency, requirements-min.txt) is not installed, so this is the strongest pin
                            of the tower math available offline.

  G9 simpler_env.json       the SimplerEnv policy wrapper OpenVLAInference.step (experiments/SimplerEnv-OpenVLA/simpler_env/
                            policies/openvla/openvla_model.py:72-145) driven by a SCRIPTED fake model: raw action sequences ->
                            (raw_action, action) dicts for both policy setups (sticky-gripper state machine, scaling, keys).
                            cv2 / transforms3d / matplotlib are absent offline: the file is imported with stub modules --
                            euler2axangle backed by scipy's Rotation (an independent implementation), cv2.resize never
                            reached (inputs are 224x224 -- the stub raises otherwise).

Usage:  python oracle/make_golden.py [g9]
"""

import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "emma-x_amd"))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from emmax.config import EmmaXConfig, LlmConfig  # noqa: E402
from emmax.tokenizer_stub import StubTokenizer  # noqa: E402
from emmax.weights import synthetic_state_dict  # noqa: E402
from oracle import emmax_oracle as orc  # noqa: E402


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def checksum(sd):
    return float(sum(float(v.double().abs().sum()) for v in sd.values()))


# ---------------------------------------------------------------------------------------------------------------------
def g1_action_decode():
    at = load(f"{REF}/prismatic/vla/action_tokenizer.py", "ref_at")
    tok = types.SimpleNamespace(vocab_size=32000)
    ref = at.ActionTokenizer(tok)
    rng = np.random.default_rng(7)
    ids = np.concatenate([np.array([31999, 31998, 31872, 31871, 31745, 31744, 31743, 30000, 5, 32000, 32063]),
                          rng.integers(31700, 32064, size=64)])
    acts = ref.decode_token_ids_to_actions(ids)
    # forward direction too: continuous -> token ids (digitize), for round-trip tests
    cont = rng.uniform(-1.2, 1.2, size=64)
    disc = np.digitize(np.clip(cont, -1.0, 1.0), ref.bins)
    np.savez(os.path.join(OUT, "action_decode.npz"), ids=ids, actions=acts, bin_centers=ref.bin_centers,
             cont=cont, cont_token_ids=32000 - disc)
    print("G1 ok", acts[:6])


def g3_projector():
    nnu = load(f"{REF}/prismatic/util/nn_utils.py", "ref_nn")
    torch.manual_seed(3)
    m = nnu.FusedMLPProjector(fused_vision_dim=34, llm_dim=64).eval()
    x = torch.randn(2, 5, 34)
    with torch.no_grad():
        y = m(x)
    sd = {k: v.numpy() for k, v in m.state_dict().items()}   # projector.{0,2,4}.{weight,bias}
    np.savez(os.path.join(OUT, "projector.npz"), x=x.numpy(), y=y.numpy(), **{k.replace(".", "__"): v for k, v in sd.items()})
    print("G3 ok", y.abs().mean().item())


def g4_llama(name, heads, kv_heads, head_dim):
    from transformers import LlamaConfig, LlamaForCausalLM

    lc = LlmConfig(hidden_size=64, intermediate_size=176, num_layers=2, num_heads=heads, num_kv_heads=kv_heads,
                   head_dim=head_dim, vocab_size=512, rms_eps=1e-5, rope_theta=10000.0)
    hf_cfg = LlamaConfig(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=heads,
                         num_key_value_heads=kv_heads, head_dim=head_dim, vocab_size=512, rms_norm_eps=1e-5,
                         rope_theta=10000.0, max_position_embeddings=2048, attention_bias=False, tie_word_embeddings=False)
    hf_cfg._attn_implementation = "eager"
    torch.manual_seed(11)
    m = LlamaForCausalLM(hf_cfg).eval().to(torch.float32)
    # widen the init so logits are not flat
    with torch.no_grad():
        for p in m.parameters():
            if p.ndim == 2:
                p.normal_(0, 0.08)
            else:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
    sd = {"language_model." + k: v.detach().clone() for k, v in m.state_dict().items()}
    T_prompt, T_new = 9, 16
    embeds = torch.randn(1, T_prompt, 64)
    with torch.no_grad():
        out = m(inputs_embeds=embeds, use_cache=True)
        prefill_logits = out.logits[0].clone()
        cache = out.past_key_values
        ids, step_logits = [], []
        nxt = int(out.logits[0, -1].argmax())
        for _ in range(T_new):
            ids.append(nxt)
            o = m(input_ids=torch.tensor([[nxt]]), past_key_values=cache, use_cache=True)
            cache = o.past_key_values
            step_logits.append(o.logits[0, -1].clone())
            nxt = int(o.logits[0, -1].argmax())
    np.savez(os.path.join(OUT, f"llama_{name}.npz"), embeds=embeds.numpy(), prefill_logits=prefill_logits.numpy(),
             step_logits=torch.stack(step_logits).numpy(), ids=np.array(ids),
             cfg=np.array([lc.hidden_size, lc.intermediate_size, lc.num_layers, lc.num_heads, lc.num_kv_heads,
                           lc.head_dim, lc.vocab_size]),
             **{k.replace(".", "__"): v.numpy() for k, v in sd.items() if "rotary" not in k})
    # self-check: the oracle must reproduce HF here, otherwise do not write a misleading fixture silently
    lg, c = orc.llama_forward(embeds, sd, lc, None)
    err = (lg[0] - prefill_logits).abs().max().item()
    print(f"G4 {name} ok; oracle-vs-HF prefill max err {err:.2e}; ids {ids[:8]}")
    assert err < 1e-4


def g5_wrapper():
    """Reference HF wrapper (splice, cache dispatch, 29871 append, de-tokenise, un-normalise) with OUR tower restatement."""
    import torch.nn as nn
    import transformers  # noqa: F401  (import before stubbing timm)
    from transformers import GenerationMixin

    cfg = EmmaXConfig.tiny()
    sd = synthetic_state_dict(cfg, seed=5, planted=True)

    class OurCpuViT(nn.Module):
        def __init__(self, tower_index):
            super().__init__()
            self.tw = cfg.towers[tower_index]
            self.prefix = orc.TOWER_PREFIXES[tower_index]
            self.blocks = [None] * self.tw.depth
            self.embed_dim = self.tw.embed_dim

        def get_intermediate_layers(self, x, n=None):
            assert n == {self.tw.depth - 2}
            return (orc.vit_tower(x.float(), sd, self.prefix, self.tw),)

    timm = types.ModuleType("timm")
    timm.__version__ = "0.9.10"
    tv = types.ModuleType("timm.models.vision_transformer")
    tm = types.ModuleType("timm.models")

    class LayerScale(nn.Module):
        pass

    tv.LayerScale = LayerScale
    tm.vision_transformer = tv
    timm.models = tm
    counter = {"i": 0}

    def create_model(name, **kw):
        i = counter["i"]
        counter["i"] += 1
        return OurCpuViT(i)

    timm.create_model = create_model
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.vision_transformer": tv})
    pkg = types.ModuleType("refhf")
    pkg.__path__ = [f"{REF}/prismatic/extern/hf"]
    sys.modules["refhf"] = pkg
    cfgm = load(f"{REF}/prismatic/extern/hf/configuration_prismatic.py", "refhf.configuration_prismatic")
    mod = load(f"{REF}/prismatic/extern/hf/modeling_prismatic.py", "refhf.modeling_prismatic")
    mod.PrismaticForConditionalGeneration.tie_weights = lambda self, *a, **k: None

    class Shim(mod.OpenVLAForActionPrediction, GenerationMixin):
        pass

    L = cfg.llm
    hcfg = cfgm.OpenVLAConfig(
        vision_backbone_id="dinosiglip-vit-so-224px", llm_backbone_id="llama2-7b-pure",
        arch_specifier="no-align+fused-gelu-mlp", image_resize_strategy="resize-naive",
        text_config=dict(hidden_size=L.hidden_size, intermediate_size=L.intermediate_size,
                         num_hidden_layers=L.num_layers, num_attention_heads=L.num_heads,
                         num_key_value_heads=L.num_kv_heads, head_dim=L.head_dim, vocab_size=L.vocab_size,
                         pad_token_id=32000, rms_norm_eps=L.rms_eps, rope_theta=L.rope_theta,
                         max_position_embeddings=2048),
        norm_stats=cfg.norm_stats)
    hcfg._attn_implementation = "eager"
    m = Shim(hcfg).eval().to(torch.float32)
    # the wrapper's projector takes vision_dim from the stub towers: must equal ours
    own = {k: v for k, v in sd.items() if k.startswith("projector.") or k.startswith("language_model.")}
    missing, unexpected = m.load_state_dict(own, strict=False)
    missing = [k for k in missing if "rotary" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    m.generation_config.eos_token_id = 2
    m.generation_config.pad_token_id = 32000
    m.generation_config.bos_token_id = 1

    rng = np.random.default_rng(1234)
    frames = rng.integers(0, 256, size=(1, 224, 224, 3), dtype=np.uint8)
    pix = orc.preprocess_frames(frames, cfg)
    from emmax.weights import planted_start_token
    prompt = [1] + [int(x) for x in rng.integers(3, 31744, size=10)] + [planted_start_token(cfg, 3)]
    ids = torch.tensor([prompt])
    with torch.no_grad():
        out = m(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=pix, use_cache=True, return_dict=True)
        logits = out.logits[0]
        gen = m.generate(ids, pixel_values=pix, attention_mask=torch.ones_like(ids), max_new_tokens=20, do_sample=False)
        ids2 = torch.tensor([prompt[:-1] + [31000]])   # predict_action appends 29871 itself
        act = m.predict_action(ids2, unnorm_key="bridge_orig", pixel_values=pix, do_sample=False)
    np.savez(os.path.join(OUT, "wrapper.npz"), frames=frames, prompt=np.array(prompt), last_logits=logits[-1].numpy(),
             logits_argmax=logits.argmax(-1).numpy(), logits_rowsum=logits.double().sum(-1).numpy(),
             generated=gen[0].numpy(), predict_prompt=ids2[0].numpy(), action=np.asarray(act),
             seed=np.array(5), weights_checksum=np.array(checksum(sd)))
    print("G5 ok; generated", gen[0, len(prompt):].tolist(), "action", np.asarray(act))
    for k in ("timm", "timm.models", "timm.models.vision_transformer"):
        sys.modules.pop(k, None)


def g6_prompts():
    pb = load(f"{REF}/prismatic/models/backbones/llm/prompting/base_prompter.py", "ref_pb")
    cases = ["What action should the robot take to achieve the instruction\nINSTRUCTION: \nPut the pot next to the cans.\n",
             "  pick up the <image> spoon  ", "x", "What action should the robot take to achieve the instruction\nINSTRUCTION: \nclose the drawer\nCURRENT GRIPPER: [48, 63]\n"]
    out = []
    for c in cases:
        b = pb.PurePromptBuilder("prismatic")
        b.add_turn("human", c)
        out.append({"message": c, "prompt": b.get_prompt(), "potential": pb.PurePromptBuilder("prismatic").get_potential_prompt(c)})
    with open(os.path.join(OUT, "prompts.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print("G6 ok")


def g7_solver():
    """Exec the reference Solver class body (solver.py:1-137 minus the hub tokenizer download at :188-190)."""
    at = load(f"{REF}/prismatic/vla/action_tokenizer.py", "ref_at2")
    src = open(f"{REF}/prismatic/vla/solver.py", encoding="utf-8").read()
    body = src[: src.index("tokenizer = AutoTokenizer.from_pretrained")]
    body = body.replace("from prismatic.vla.action_tokenizer import ActionTokenizer", "")
    body = body.replace("from transformers import AutoTokenizer", "")
    ns = {"ActionTokenizer": at.ActionTokenizer}
    exec(compile(body, "ref_solver", "exec"), ns)
    tok = StubTokenizer()
    solver = ns["Solver"](at.ActionTokenizer(tok), verbose=False)

    def act_text(ids):
        return tok.decode(ids)

    a1 = [31900, 31800, 31850, 31760, 31990, 31872, 31871, 31745]
    a2 = [31901, 31744, 31999, 31800, 31801, 31802, 31803, 31804]
    cases = [
        "REASONING:\nmove it.\nSUBTASK: lift\n\nNEXT GRIPPER: [105, 74]\n\nMOVEMENT:\n" + act_text(a2) + "\nPOLICIES:\n" + act_text(a1) + ";" + act_text(a2) + "\n",
        "MOVEMENT:\n" + act_text(a2) + "\nPOLICIES:\n" + act_text(a1) + "\n",
        act_text(a1),                                      # no key -> whole text is the policy
        "POLICIES:\n" + act_text(a1[:5]) + "\n",           # wrong length -> reference's exception path -> [[0]*7]
        "POLICIES:\n\n\n",                                 # empty -> exception path
        "no actions here",
        "MOVEMENT:\nmove forward 3; move left 2; rotate clockwise 10; close gripper\nPOLICIES:\n" + act_text(a1),
        "MOVEMENT:\nmove sideways 3; close gripper\n",     # unknown direction -> [-100]*7
        # extract_2d_coordinates (solver.py:33-40 evals the line): tuple, floats, arithmetic, nesting, garbage, missing line
        "NEXT GRIPPER: (105, 74)\n",
        "NEXT GRIPPER:\n\n  [105.5, 74]  \nMOVEMENT:\n",
        "NEXT GRIPPER: 105, 74\n",
        "NEXT GRIPPER: [100 + 5, 148 // 2]\n",
        "NEXT GRIPPER: [[1, 2], [3, 4]]\n",
        "NEXT GRIPPER: -3.5\n",
        "NEXT GRIPPER: one hundred, five\n",
        "NEXT GRIPPER: [105, 74\n",
        "NEXT GRIPPER: [105, x]\n",                        # undefined name -> NameError -> [0, 0]
        "NEXT GRIPPER:\n",
    ]
    out = []
    for c in cases:
        pol, remain = solver.extract_action_policies(c)
        req, mv = solver.extract_movement_plan(c)
        coord = solver.extract_2d_coordinates(c)
        out.append({"text": c, "policies": pol, "remain": remain, "require_unorm": req, "movement": np.asarray(mv).tolist(),
                    "coordinates": json.loads(json.dumps(coord)), "coordinates_type": type(coord).__name__})
    with open(os.path.join(OUT, "solver.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print("G7 ok", [len(o["policies"]) for o in out])


VIT_SEED, VIT_INPUT_SEED = 31, 77


def vit_golden_inputs(cfg):
    """Deterministic pixel input of the G8 vectors (also used by the test): [1, 3, 224, 224] per tower, normalised scale."""
    rng = np.random.default_rng(VIT_INPUT_SEED)
    return [torch.from_numpy(rng.standard_normal((1, 3, tw.image_size, tw.image_size)).astype(np.float32)) for tw in cfg.towers]


def g8_vit_hf():
    from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel, SiglipVisionConfig, SiglipVisionModel

    cfg = EmmaXConfig.tiny()
    sd = synthetic_state_dict(cfg, seed=VIT_SEED)
    xs = vit_golden_inputs(cfg)
    out = {"seed": np.int64(VIT_SEED), "input_seed": np.int64(VIT_INPUT_SEED), "checksum": np.float64(checksum(sd))}
    for ti, (pre, tw) in enumerate(zip(orc.TOWER_PREFIXES, cfg.towers)):
        g = lambda k: sd[pre + k].float()
        D = tw.embed_dim
        if tw.has_cls:   # DINOv2 with 4 registers + LayerScale
            hc = Dinov2WithRegistersConfig(hidden_size=D, num_hidden_layers=tw.depth, num_attention_heads=tw.num_heads,
                                           mlp_ratio=tw.mlp_hidden // D, image_size=tw.image_size, patch_size=tw.patch,
                                           num_register_tokens=tw.n_reg, layer_norm_eps=tw.ln_eps, hidden_act="gelu",
                                           use_swiglu_ffn=False, qkv_bias=True)
            assert tw.mlp_hidden == hc.mlp_ratio * D
            m = Dinov2WithRegistersModel(hc).eval()
            t = {"embeddings.cls_token": g("cls_token"), "embeddings.register_tokens": g("reg_token"),
                 "embeddings.mask_token": torch.zeros(1, D),
                 # timm reg4 models: no_embed_class=True -> the class token gets no position embedding
                 "embeddings.position_embeddings": torch.cat([torch.zeros(1, 1, D), g("pos_embed")], dim=1),
                 "embeddings.patch_embeddings.projection.weight": g("patch_embed.proj.weight"),
                 "embeddings.patch_embeddings.projection.bias": g("patch_embed.proj.bias"),
                 "layernorm.weight": torch.ones(D), "layernorm.bias": torch.zeros(D)}
            for i in range(tw.depth):
                p, q = f"blocks.{i}.", f"encoder.layer.{i}."
                wq, wk, wv = g(p + "attn.qkv.weight").chunk(3, dim=0)
                bq, bk, bv = g(p + "attn.qkv.bias").chunk(3, dim=0)
                t.update({q + "norm1.weight": g(p + "norm1.weight"), q + "norm1.bias": g(p + "norm1.bias"),
                          q + "attention.attention.query.weight": wq, q + "attention.attention.query.bias": bq,
                          q + "attention.attention.key.weight": wk, q + "attention.attention.key.bias": bk,
                          q + "attention.attention.value.weight": wv, q + "attention.attention.value.bias": bv,
                          q + "attention.output.dense.weight": g(p + "attn.proj.weight"), q + "attention.output.dense.bias": g(p + "attn.proj.bias"),
                          q + "layer_scale1.lambda1": g(p + "ls1.scale_factor"), q + "layer_scale2.lambda1": g(p + "ls2.scale_factor"),
                          q + "norm2.weight": g(p + "norm2.weight"), q + "norm2.bias": g(p + "norm2.bias"),
                          q + "mlp.fc1.weight": g(p + "mlp.fc1.weight"), q + "mlp.fc1.bias": g(p + "mlp.fc1.bias"),
                          q + "mlp.fc2.weight": g(p + "mlp.fc2.weight"), q + "mlp.fc2.bias": g(p + "mlp.fc2.bias")})
            missing, unexpected = m.load_state_dict(t, strict=False)
            assert not unexpected and not missing, (missing, unexpected)
        else:            # SigLIP vision tower (no class token, learned absolute positions)
            hc = SiglipVisionConfig(hidden_size=D, intermediate_size=tw.mlp_hidden, num_hidden_layers=tw.depth,
                                    num_attention_heads=tw.num_heads, image_size=tw.image_size, patch_size=tw.patch,
                                    hidden_act="gelu", layer_norm_eps=tw.ln_eps)
            m = SiglipVisionModel(hc).eval()
            keys = list(m.state_dict().keys())
            root = "vision_model." if any(k.startswith("vision_model.") for k in keys) else ""
            t = {root + "embeddings.patch_embedding.weight": g("patch_embed.proj.weight"),
                 root + "embeddings.patch_embedding.bias": g("patch_embed.proj.bias"),
                 root + "embeddings.position_embedding.weight": g("pos_embed")[0]}
            for i in range(tw.depth):
                p, q = f"blocks.{i}.", root + f"encoder.layers.{i}."
                wq, wk, wv = g(p + "attn.qkv.weight").chunk(3, dim=0)
                bq, bk, bv = g(p + "attn.qkv.bias").chunk(3, dim=0)
                t.update({q + "layer_norm1.weight": g(p + "norm1.weight"), q + "layer_norm1.bias": g(p + "norm1.bias"),
                          q + "self_attn.q_proj.weight": wq, q + "self_attn.q_proj.bias": bq,
                          q + "self_attn.k_proj.weight": wk, q + "self_attn.k_proj.bias": bk,
                          q + "self_attn.v_proj.weight": wv, q + "self_attn.v_proj.bias": bv,
                          q + "self_attn.out_proj.weight": g(p + "attn.proj.weight"), q + "self_attn.out_proj.bias": g(p + "attn.proj.bias"),
                          q + "layer_norm2.weight": g(p + "norm2.weight"), q + "layer_norm2.bias": g(p + "norm2.bias"),
                          q + "mlp.fc1.weight": g(p + "mlp.fc1.weight"), q + "mlp.fc1.bias": g(p + "mlp.fc1.bias"),
                          q + "mlp.fc2.weight": g(p + "mlp.fc2.weight"), q + "mlp.fc2.bias": g(p + "mlp.fc2.bias")})
            missing, unexpected = m.load_state_dict(t, strict=False)   # post_layernorm / pooling head are not on the path
            assert not unexpected and all(("post_layernorm" in k or "head." in k) for k in missing), (missing, unexpected)
        with torch.no_grad():
            hs = m(pixel_values=xs[ti], output_hidden_states=True).hidden_states
        assert len(hs) == tw.depth + 1
        ref = hs[tw.take_index + 1][:, tw.n_prefix:, :]          # after block take_index, prefix tokens dropped, no final norm
        ours = orc.vit_tower(xs[ti], sd, pre, tw)
        err = float((ours - ref).abs().max() / ref.abs().max())
        assert ref.shape == (1, tw.n_patches, D) and err < 1e-4, (ref.shape, err)
        out[f"tower{ti}_expected"] = ref.numpy().astype(np.float32)
        print(f"G8 tower {ti} ({type(m).__name__}) ok, oracle rel err {err:.2e}")
    np.savez_compressed(os.path.join(OUT, "vit_hf.npz"), **out)


def g9_simpler_env():
    from scipy.spatial.transform import Rotation

    def euler2axangle(ai, aj, ak):
        rv = Rotation.from_euler("xyz", [ai, aj, ak]).as_rotvec()
        ang = float(np.linalg.norm(rv))
        return (rv / ang, ang) if ang > 0 else (np.array([1.0, 0.0, 0.0]), 0.0)

    def no_resize(image, size, interpolation=None):
        assert tuple(image.shape[:2]) == (size[1], size[0]), "golden inputs are already at image_size"
        return image

    stubs = {"cv2": types.SimpleNamespace(resize=no_resize, INTER_AREA=3), "matplotlib": types.ModuleType("matplotlib"),
             "matplotlib.pyplot": types.ModuleType("matplotlib.pyplot"), "transforms3d": types.ModuleType("transforms3d"),
             "transforms3d.euler": types.SimpleNamespace(euler2axangle=euler2axangle)}
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    import transformers
    had_v2s = hasattr(transformers, "AutoModelForVision2Seq")
    if not had_v2s:   # transformers >= 5 dropped the name the reference imports (never called here: the constructor is bypassed)
        transformers.AutoModelForVision2Seq = transformers.AutoModelForImageTextToText
    try:
        mod = load(f"{REF}/experiments/SimplerEnv-OpenVLA/simpler_env/policies/openvla/openvla_model.py", "ref_simpler")
    finally:
        if not had_v2s:
            del transformers.AutoModelForVision2Seq
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v

    class FakeVLA:
        def __init__(self, seq):
            self.seq, self.i, self.calls = seq, 0, []

        def predict_action(self, **kw):
            self.calls.append(kw.get("unnorm_key"))
            a = np.asarray(self.seq[self.i], dtype=np.float64)
            self.i += 1
            return a

    class FakeInputs(dict):
        def to(self, *a, **k):
            return self

    rng = np.random.default_rng(5)
    grip = [1.0, 0.9, 0.1, 0.05, 0.0, 0.0, 0.95, 1.0, 0.2, 0.2, 0.9, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.8, 0.8]
    seq = [list(rng.uniform(-0.05, 0.05, 3)) + list(rng.uniform(-0.3, 0.3, 3)) + [g] for g in grip]
    seq[3][3:6] = [0.0, 0.0, 0.0]                         # null rotation
    out = {"raw_sequence": seq, "runs": []}
    for setup, scale in (("widowx_bridge", 1.0), ("google_robot", 1.0), ("google_robot", 0.5)):
        pol = mod.OpenVLAInference.__new__(mod.OpenVLAInference)   # the constructor downloads from the hub: set its state by hand
        pol.policy_setup, pol.unnorm_key = setup, {"widowx_bridge": "bridge_orig", "google_robot": "fractal20220817_data"}[setup]
        pol.sticky_gripper_num_repeat = {"widowx_bridge": 1, "google_robot": 15}[setup]
        pol.image_size, pol.action_scale = [224, 224], scale
        pol.horizon = pol.pred_action_horizon = pol.exec_horizon = 1
        pol.task = pol.task_description = None
        pol.num_image_history = 0
        pol.sticky_action_is_on, pol.gripper_action_repeat, pol.sticky_gripper_action, pol.previous_gripper_action = False, 0, 0.0, None
        pol.vla = FakeVLA(seq)
        pol.processor = lambda prompt, image: FakeInputs()
        steps = []
        img = np.zeros((224, 224, 3), dtype=np.uint8)
        for i in range(len(seq)):
            task = "pick up the spoon" if i < 20 else "close the drawer"      # a new description resets the policy state
            raw, act = pol.step(img, task)
            steps.append({"task": task, "raw": {k: np.asarray(v).tolist() for k, v in raw.items()},
                          "action": {k: np.asarray(v, dtype=np.float64).tolist() for k, v in act.items()}})
        out["runs"].append({"policy_setup": setup, "action_scale": scale, "unnorm_keys_seen": sorted(set(pol.vla.calls)), "steps": steps})
    with open(os.path.join(OUT, "simpler_env.json"), "w") as f:
        json.dump(out, f)
    print("G9 ok", [len(r["steps"]) for r in out["runs"]])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ["g9"]:
        g9_simpler_env()
        sys.exit(0)
    g1_action_decode()
    g3_projector()
    g4_llama("mha", 4, 4, 16)
    g4_llama("gqa", 4, 2, 32)
    g6_prompts()
    g7_solver()
    g5_wrapper()
    g8_vit_hf()
    g9_simpler_env()
