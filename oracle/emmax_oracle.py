"""
emmax_oracle.py -- CPU restatement of the Emma-X VLA forward/generate hot path.

*** TEST INFRASTRUCTURE ONLY. ***  Nothing under `emma-x_amd/` may import, call, link or execute this file. Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` use it, and there only as the checker /
the timed CPU baseline -- never as the thing shipped.

What it restates (all `file:line` relative to /root/reference):
  * image normalisation              prismatic/extern/hf/processing_prismatic.py:128-145
  * fused DINOv2+SigLIP backbone     prismatic/extern/hf/modeling_prismatic.py:114-123 (+ timm 0.9.10 ViT math)
  * FusedMLPProjector                prismatic/util/nn_utils.py:37-53, modeling_prismatic.py:146-158
  * multimodal splice + prefill      modeling_prismatic.py:362-415
  * cached decode step               modeling_prismatic.py:325-341
  * LLaMA-2 decoder math             transformers `LlamaForCausalLM` (pinned 4.40.1 in requirements-min.txt:5; not vendored)
  * greedy generation                transformers `GenerationMixin.generate` invoked at modeling_prismatic.py:519,
                                     prismatic/models/vlms/prismatic.py:659-663
  * action de-tokenisation           prismatic/vla/action_tokenizer.py:49-68, modeling_prismatic.py:522-535
  * Solver text -> actions           prismatic/vla/solver.py:42-137
  * PurePromptBuilder                prismatic/models/backbones/llm/prompting/base_prompter.py:28-73

Pinning status (see DESIGN.md "Oracle"):
  * LLaMA math, greedy loop  -> pinned against HF `LlamaForCausalLM` CPU fp32 (tests/golden/llama_*.npz, made by
                                 oracle/make_golden.py in the authoring container).
  * splice / cache dispatch / predict_action tail -> pinned against the reference's own
                                 `OpenVLAForActionPrediction` run through a stub-timm shim (tests/golden/wrapper_*.npz).
  * action tokenizer, projector, prompt builder, un-normalisation -> pinned against the reference files imported by path.
  * ViT towers (timm 0.9.10)  -> pinned against an INDEPENDENT implementation of the same published architectures:
                                 transformers' Dinov2WithRegistersModel and SiglipVisionModel (exact-erf GELU) on seeded
                                 weights (tests/golden/vit_hf.npz, agreement 1e-7).  **Parity with timm itself stays
                                 unpinned**: timm is neither vendored in the reference nor installed here, so the points where
                                 timm 0.9.10 could differ from HF (activation of the SigLIP MLP, `no_embed_class`, which block
                                 `get_intermediate_layers` returns) follow SURVEY.md Appendix B.
  * tokenizer text<->ids round trips -> **parity unpinned** (no LLaMA tokenizer files offline); goldens operate on ids.

Everything is plain PyTorch on CPU. `dtype=torch.float32` is the reference's CPU path (BASELINE config 1);
`dtype=torch.bfloat16` executes the same op graph the way the reference would on an accelerator in bf16 (each torch op
rounds its output to bf16) and is used to size the tolerance of the bf16 HIP path.
"""

from __future__ import annotations

import math
from collections import defaultdict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# =====================================================================================================================
# Image pre-processing (host side of the path)
# =====================================================================================================================
def preprocess_frames(frames_u8: np.ndarray, cfg) -> Tensor:
    """uint8 [B,224,224,3] RGB -> f32 [B,6,224,224]; channels 0-2 = tower-0 normalisation, 3-5 = tower-1.

    processing_prismatic.py:128-145 with `resize-naive` at native 224x224: resize and center-crop are identities, so the
    transform is to_tensor (/255) followed by per-tower (x-mean)/std, then a channel stack (`torch.vstack`).
    """
    x = torch.from_numpy(np.ascontiguousarray(frames_u8)).permute(0, 3, 1, 2).to(torch.float32) / 255.0
    outs = []
    for tw in cfg.towers:
        mean = torch.tensor(tw.mean, dtype=torch.float32).view(1, 3, 1, 1)
        std = torch.tensor(tw.std, dtype=torch.float32).view(1, 3, 1, 1)
        outs.append((x - mean) / std)
    return torch.cat(outs, dim=1)


# =====================================================================================================================
# ViT towers (timm 0.9.10 VisionTransformer semantics) -- pinned against HF Dinov2WithRegisters /
# SiglipVision (tests/golden/vit_hf.npz); parity with timm itself unpinned
# =====================================================================================================================
def vit_tower(x: Tensor, sd: Dict[str, Tensor], prefix: str, tw, dtype=torch.float32, n_blocks: Optional[int] = None) -> Tensor:
    """One tower: x [B,3,224,224] -> patch tokens of block `take_index` [B,256,D], no final norm.

    modeling_prismatic.py:85-87 hard-wires `get_intermediate_layers(n={depth-2})`; timm returns the output of that block
    with prefix (cls/reg) tokens dropped and `norm=False`.  Blocks after `take_index` do not influence the result.
    `n_blocks` (bench only) truncates the tower to its first n blocks for the bounded CPU-baseline sample.
    """
    g = lambda k: sd[prefix + k].to(dtype)
    x = x.to(dtype)
    B = x.shape[0]
    D, H = tw.embed_dim, tw.num_heads
    hd = D // H
    t = F.conv2d(x, g("patch_embed.proj.weight"), g("patch_embed.proj.bias"), stride=tw.patch)  # [B,D,16,16]
    t = t.flatten(2).transpose(1, 2)  # [B,256,D]
    t = t + g("pos_embed")  # DINOv2 reg4: no_embed_class=True -> pos on patch tokens only
    if tw.n_prefix > 0:
        pre = [g("cls_token").expand(B, -1, -1)]
        if tw.n_reg > 0:
            pre.append(g("reg_token").expand(B, -1, -1))
        t = torch.cat(pre + [t], dim=1)
    last = tw.take_index if n_blocks is None else min(n_blocks - 1, tw.take_index)
    for i in range(last + 1):
        p = f"blocks.{i}."
        h = F.layer_norm(t, (D,), g(p + "norm1.weight"), g(p + "norm1.bias"), eps=tw.ln_eps)
        qkv = F.linear(h, g(p + "attn.qkv.weight"), g(p + "attn.qkv.bias"))
        N = qkv.shape[1]
        qkv = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        att = torch.matmul(q * (hd ** -0.5), k.transpose(-2, -1))
        att = att.softmax(dim=-1)
        a = torch.matmul(att, v).transpose(1, 2).reshape(B, N, D)
        a = F.linear(a, g(p + "attn.proj.weight"), g(p + "attn.proj.bias"))
        if tw.layerscale:
            a = a * g(p + "ls1.scale_factor")
        t = t + a
        h = F.layer_norm(t, (D,), g(p + "norm2.weight"), g(p + "norm2.bias"), eps=tw.ln_eps)
        f = F.gelu(F.linear(h, g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias")))  # exact-erf GELU
        f = F.linear(f, g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias"))
        if tw.layerscale:
            f = f * g(p + "ls2.scale_factor")
        t = t + f
    return t[:, tw.n_prefix:, :]


TOWER_PREFIXES = ("vision_backbone.featurizer.", "vision_backbone.fused_featurizer.")


def vision_backbone(pixel_values: Tensor, sd, cfg, dtype=torch.float32, n_blocks=None) -> Tensor:
    """[B,6,224,224] -> [B,256,D0+D1]  (modeling_prismatic.py:114-123: split [3,3], featurize, cat dim=2)."""
    feats = []
    for i, tw in enumerate(cfg.towers):
        feats.append(vit_tower(pixel_values[:, 3 * i:3 * i + 3], sd, TOWER_PREFIXES[i], tw, dtype, n_blocks))
    return torch.cat(feats, dim=2)


def projector(x: Tensor, sd, dtype=torch.float32) -> Tensor:
    """fc3(gelu(fc2(gelu(fc1(x)))))  (nn_utils.py:37-53 / modeling_prismatic.py:151-156)."""
    g = lambda k: sd["projector." + k].to(dtype)
    x = x.to(dtype)
    x = F.gelu(F.linear(x, g("fc1.weight"), g("fc1.bias")))
    x = F.gelu(F.linear(x, g("fc2.weight"), g("fc2.bias")))
    return F.linear(x, g("fc3.weight"), g("fc3.bias"))


# =====================================================================================================================
# LLaMA-2 decoder (HF LlamaForCausalLM semantics)
# =====================================================================================================================
def rms_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """HF LlamaRMSNorm: upcast fp32, x*rsqrt(mean(x^2)+eps), downcast, *weight."""
    dt = x.dtype
    xf = x.to(torch.float32)
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(dt)


def rope_cos_sin(positions: Tensor, head_dim: int, theta: float, dtype) -> Tuple[Tensor, Tensor]:
    """HF LlamaRotaryEmbedding: inv_freq = theta^(-2i/d) (fp32), emb = cat(freqs, freqs), cos/sin fp32 -> act dtype."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64, device=positions.device).to(torch.float32) / head_dim))
    freqs = positions.to(torch.float32)[:, None] * inv_freq[None, :]
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rotate_half(x: Tensor) -> Tensor:
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def llama_layer(h: Tensor, sd, li: int, lc, positions: Tensor, kv: Optional[Tuple[Tensor, Tensor]], dtype):
    """One decoder layer on h [B,T,hidden] at absolute `positions` [T]; kv = cached (K,V) [B,Hkv,L,hd] or None.

    Attention is causal over [cache ++ new] (HF eager: fp32 softmax, then cast to activation dtype).
    Returns (h_out, (K_all, V_all)).
    """
    p = f"language_model.model.layers.{li}."
    g = lambda k: sd[p + k].to(dtype)
    B, T, _ = h.shape
    Hq, Hkv, hd = lc.num_heads, lc.num_kv_heads, lc.head_dim
    x = rms_norm(h, g("input_layernorm.weight"), lc.rms_eps)
    q = F.linear(x, g("self_attn.q_proj.weight")).view(B, T, Hq, hd).transpose(1, 2)
    k = F.linear(x, g("self_attn.k_proj.weight")).view(B, T, Hkv, hd).transpose(1, 2)
    v = F.linear(x, g("self_attn.v_proj.weight")).view(B, T, Hkv, hd).transpose(1, 2)
    cos, sin = rope_cos_sin(positions, hd, lc.rope_theta, dtype)
    cos, sin = cos[None, None], sin[None, None]
    q = q * cos + _rotate_half(q) * sin
    k = k * cos + _rotate_half(k) * sin
    if kv is not None:
        k = torch.cat([kv[0], k], dim=2)
        v = torch.cat([kv[1], v], dim=2)
    L = k.shape[2]
    rep = Hq // Hkv
    kk = k.repeat_interleave(rep, dim=1) if rep > 1 else k
    vv = v.repeat_interleave(rep, dim=1) if rep > 1 else v
    att = torch.matmul(q, kk.transpose(2, 3)) * (hd ** -0.5)
    # causal mask: query at absolute position positions[t] may see keys 0..positions[t]
    key_pos = torch.arange(L, device=positions.device)
    mask = key_pos[None, :] > positions[:, None]
    att = att.masked_fill(mask[None, None], float("-inf"))
    att = F.softmax(att, dim=-1, dtype=torch.float32).to(dtype)
    a = torch.matmul(att, vv).transpose(1, 2).reshape(B, T, Hq * hd)
    h = h + F.linear(a, g("self_attn.o_proj.weight"))
    x = rms_norm(h, g("post_attention_layernorm.weight"), lc.rms_eps)
    m = F.linear(F.silu(F.linear(x, g("mlp.gate_proj.weight"))) * F.linear(x, g("mlp.up_proj.weight")),
                 g("mlp.down_proj.weight"))
    return h + m, (k, v)


def llama_forward(embeds: Tensor, sd, lc, kv_cache: Optional[List], dtype=torch.float32,
                  n_layers: Optional[int] = None, last_only: bool = False):
    """Run the decoder stack on `embeds` [B,T,hidden] appended after the cache. Returns (logits [B,T|1,V], new_cache)."""
    h = embeds.to(dtype)
    T = h.shape[1]
    past = 0 if kv_cache is None else kv_cache[0][0].shape[2]
    positions = torch.arange(past, past + T, device=h.device)   # tensors follow the inputs' device (tests may execute the fp32 restatement on the GPU)
    new_cache = []
    nl = lc.num_layers if n_layers is None else n_layers
    for li in range(nl):
        h, kv = llama_layer(h, sd, li, lc, positions, None if kv_cache is None else kv_cache[li], dtype)
        new_cache.append(kv)
    if last_only:
        h = h[:, -1:, :]
    h = rms_norm(h, sd["language_model.model.norm.weight"].to(dtype), lc.rms_eps)
    logits = F.linear(h, sd["language_model.lm_head.weight"].to(dtype))
    return logits, new_cache


def embed_tokens(ids: Tensor, sd, dtype=torch.float32) -> Tensor:
    return F.embedding(ids.long(), sd["language_model.model.embed_tokens.weight"]).to(dtype)


def splice(input_ids: Tensor, patch_embeds: Tensor, sd, dtype=torch.float32) -> Tensor:
    """[BOS] ++ patches ++ text[1:]  (modeling_prismatic.py:380-385)."""
    e = embed_tokens(input_ids, sd, dtype)
    return torch.cat([e[:, :1], patch_embeds.to(dtype), e[:, 1:]], dim=1)


# =====================================================================================================================
# Full path: frames + prompt ids -> generated ids  (batch-size-1 semantics per row; Appendix C of SURVEY.md)
# =====================================================================================================================
def vla_prefill_logits(input_ids: Tensor, pixel_values: Tensor, sd, cfg, dtype=torch.float32):
    """Multimodal forward: returns (logits [B,S,V], kv_cache, projected patches)."""
    patches = vision_backbone(pixel_values, sd, cfg, dtype)
    proj = projector(patches, sd, dtype)
    emb = splice(input_ids, proj, sd, dtype)
    logits, cache = llama_forward(emb, sd, cfg.llm, None, dtype)
    return logits, cache, proj


def greedy_generate(input_ids: Tensor, pixel_values: Tensor, sd, cfg, max_new_tokens: int, dtype=torch.float32,
                    eos_token_id: Optional[int] = 2, return_trace: bool = False):
    """HF greedy search for ONE sequence (the reference refuses B>1: modeling_prismatic.py:460-463).

    Step 0 = multimodal prefill, steps t>=1 = cached one-token decode; next = argmax(logits[:, -1]) (first max wins);
    stop when next == eos (the eos token IS appended, as HF does) or after max_new_tokens.
    Returns ids [1,P+T] (and, with return_trace, per-step last-position logits f32 [T,V]).
    """
    assert input_ids.shape[0] == 1
    logits, cache, _ = vla_prefill_logits(input_ids, pixel_values, sd, cfg, dtype)
    out = [int(t) for t in input_ids[0]]
    trace = []
    for _ in range(max_new_tokens):
        last = logits[0, -1].to(torch.float32)
        if return_trace:
            trace.append(last.clone())
        nxt = int(torch.argmax(last))
        out.append(nxt)
        if eos_token_id is not None and nxt == eos_token_id:
            break
        e = embed_tokens(torch.tensor([[nxt]]), sd, dtype)
        logits, cache = llama_forward(e, sd, cfg.llm, cache, dtype)
    ids = torch.tensor([out], dtype=torch.long)
    if return_trace:
        return ids, torch.stack(trace)
    return ids


# =====================================================================================================================
# Action de-tokenisation / un-normalisation / Solver / prompt builder (host integer + fp64 math)
# =====================================================================================================================
def bin_centers(n_bins: int = 256) -> np.ndarray:
    bins = np.linspace(-1, 1, n_bins)  # action_tokenizer.py:30-31
    return (bins[:-1] + bins[1:]) / 2.0


def decode_token_ids_to_actions(ids: np.ndarray, vocab_size: int = 32000, n_bins: int = 256) -> np.ndarray:
    """action_tokenizer.py:49-68 / modeling_prismatic.py:523-525:  d = V - id; idx = clip(d-1, 0, n_bins-2)."""
    c = bin_centers(n_bins)
    d = vocab_size - np.asarray(ids)
    d = np.clip(d - 1, a_min=0, a_max=c.shape[0] - 1)
    return c[d]


def unnormalize_actions(a: np.ndarray, stats: dict) -> np.ndarray:
    """modeling_prismatic.py:528-535."""
    mask = stats.get("mask", np.ones_like(stats["q01"], dtype=bool))
    hi, lo = np.array(stats["q99"]), np.array(stats["q01"])
    return np.where(mask, 0.5 * (np.asarray(a) + 1) * (hi - lo) + lo, a)


def predict_action_tail(generated_ids: np.ndarray, stats: dict, vocab_size: int = 32000, n_bins: int = 256) -> np.ndarray:
    """modeling_prismatic.py:522-535 applied to the last `action_dim` generated ids."""
    dim = len(stats["q01"])
    return unnormalize_actions(decode_token_ids_to_actions(generated_ids[-dim:], vocab_size, n_bins), stats)


def generate_actions_pos_tail(require_unorm, delta_position, proprio_stats: dict):
    """prismatic/models/vlms/prismatic.py:686-696, the `type == "pos"` tail of `generate_actions`, branch for branch -- INCLUDING its
    defect: `proprio_norm` is only assigned under `if require_unorm:`, so a textual movement line (require_unorm False, solver.py:57-58)
    or an unparsable one (require_unorm None, solver.py:43 / the catch-all) ends in the reference's `UnboundLocalError`.  The product
    (emmax/modeling.py `_postprocess`) returns the Solver's delta unchanged on that branch instead."""
    if require_unorm:
        mask = proprio_stats.get("mask", np.ones_like(proprio_stats["Q1"], dtype=bool))
        hi, lo = np.array(proprio_stats["Q99"]), np.array(proprio_stats["Q1"])
        proprio_norm = np.where(mask, 0.5 * (np.array(delta_position) + 1) * (hi - lo) + lo, delta_position)
    return proprio_norm   # noqa: F821 -- unbound when require_unorm is falsy, as in the reference


class Solver:
    """Restatement of prismatic/vla/solver.py:8-137 (`tokenizer` = object with __call__(text, add_special_tokens) ->
    .input_ids; never raises: parse failures yield zeros / -100s exactly as the reference)."""

    def __init__(self, tokenizer, vocab_size: int = 32000, n_bins: int = 256):
        self.tok, self.vocab_size, self.n_bins = tokenizer, vocab_size, n_bins
        self.movement_key, self.policy_key, self.coordinates_key = "MOVEMENT:", "POLICIES:", "NEXT GRIPPER:"

    def extract_2d_coordinates(self, text: str):
        """prismatic/vla/solver.py:33-40: first non-empty line after the key, `eval`ed as the reference does (test inputs only)."""
        try:
            at = text.index(self.coordinates_key) + len(self.coordinates_key)
            lines = [o for o in text[at:].split("\n") if len(o.strip()) != 0]
            return eval(lines[0].strip())
        except Exception:
            return [0, 0]

    def _decode(self, ids):
        return decode_token_ids_to_actions(np.array(ids), self.vocab_size, self.n_bins)

    def extract_action_policies(self, text: str):
        try:
            if self.policy_key in text:
                idx = text.index(self.policy_key) + len(self.policy_key)
                policy = text[idx:]
                remain = text[: text.index(self.policy_key)]
                policies = [o for o in policy.split("\n") if len(o.strip()) != 0]
                policies = policies[0].strip()
            else:
                policies = text.strip()
                remain = ""
            out = []
            for piece in policies.split(";"):
                tok = self.tok(piece, add_special_tokens=False).input_ids
                a = self._decode(tok)
                a = a[1:]   # "the first token is meaningless" (solver.py:125-126)
                a = a[:7]
                if len(a) != 7:
                    a = [0] * 7
                out.append(a.tolist())   # NB: reference calls .tolist() on the python list too -> exception path
        except Exception:
            out = [[0] * 7]
            remain = text
        return out, remain

    def extract_movement_plan(self, text: str):
        require_unorm = None
        try:
            idx = text.index(self.movement_key) + len(self.movement_key)
            lvl = [o for o in text[idx:].split("\n") if len(o.strip()) != 0][0].strip()
            if "gripper" not in lvl:
                require_unorm = True
                ids = self.tok(lvl, add_special_tokens=False).input_ids
                mv = self._decode(ids)[1:8]
                assert len(mv) == 7
            else:
                require_unorm = False
                parts = [o for o in lvl.split(";") if len(o) > 0][:7]
                pos = defaultdict(int)
                table = dict(move_backward=(-1, "y"), move_forward=(1, "y"), move_right=(-1, "x"), move_left=(1, "x"),
                             move_downward=(-1, "z"), move_upward=(1, "z"), roll_downward=(-1, "ox"),
                             roll_upward=(1, "ox"), swing_downward=(-1, "ox"), swing_upward=(1, "ox"),
                             pitch_downward=(-1, "oy"), pitch_upward=(1, "oy"), yaw_downward=(-1, "oz"),
                             yaw_upward=(1, "oz"), rotate_clockwise=(-1, "oz"), rotate_counterclockwise=(1, "oz"),
                             close_gripper=(-1, "grip"), open_gripper=(1, "grip"))
                for ml in parts:
                    sign, axis = table["_".join(ml.split()[:2])]
                    scale = 1
                    if "o" in axis:
                        scale = scale * 1e-3
                    elif "grip" in axis:
                        scale = scale
                    else:
                        scale = scale / 180 * np.pi
                    level = round("open" in ml) if "grip" in axis else int(ml.split()[2])
                    pos[axis] += sign * scale * level
                mv = [pos[k] for k in ["x", "y", "z", "ox", "oy", "oz", "grip"]]
        except Exception:
            mv = [-100] * 7
        return require_unorm, np.array(mv)


def pure_prompt(message: str) -> str:
    """PurePromptBuilder single human turn + get_prompt()  (base_prompter.py:36,42-43,71-73)."""
    msg = message.replace("<image>", "").strip()
    return f"In: {msg}\nOut: ".removeprefix("<s>").rstrip()


def bridge_task_prompt(task: str, gripper_xy: Optional[Sequence[int]] = None) -> str:
    """experiments/robot/bridge/run_bridgev2_eval.py:167-168 template."""
    s = f"What action should the robot take to achieve the instruction\nINSTRUCTION: \n{task}\n"
    if gripper_xy is not None:
        s += f"CURRENT GRIPPER: [{gripper_xy[0]}, {gripper_xy[1]}]\n"
    return s
