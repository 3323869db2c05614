"""
weights.py -- state-dict contract of the hot path: key names/shapes, a seeded synthetic generator, and the HF
safetensors / native `.pt` readers.

The on-disk layout is the one the reference's converter writes (vla-scripts/extern/convert_openvla_weights_to_hf.py:74-116):
  projector.fc{1,2,3}.{weight,bias}
  language_model.model.embed_tokens.weight, language_model.model.layers.{i}.{self_attn.{q,k,v,o}_proj,mlp.{gate,up,down}_proj}.weight,
  language_model.model.layers.{i}.{input,post_attention}_layernorm.weight, language_model.model.norm.weight, language_model.lm_head.weight
  vision_backbone.featurizer.*        (DINOv2: cls_token, reg_token, pos_embed, patch_embed.proj, blocks.{i}.*, ls{1,2}.scale_factor)
  vision_backbone.fused_featurizer.*  (SigLIP: pos_embed, patch_embed.proj, blocks.{i}.*)
The native Prismatic `.pt` layout ({"model": {"vision_backbone", "projector", "llm_backbone"}}, prismatic/models/vlms/prismatic.py:112-120)
is remapped to the same keys by `remap_native_state_dict` (mirrors convert_openvla_weights_to_hf.py:84-116).
No real checkpoint exists offline; `synthetic_state_dict` produces tensors with the real names and shapes.
"""

from __future__ import annotations

import glob
import json
import os
from typing import Dict, Iterator, List, Optional, Tuple

import torch

from .config import EmmaXConfig, TowerConfig

TOWER_PREFIXES = ("vision_backbone.featurizer.", "vision_backbone.fused_featurizer.")


def tower_param_shapes(tw: TowerConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key-suffix, shape, kind) for every tensor of one tower the path reads. kind in {w, b, one, ls, tok}."""
    D, M = tw.embed_dim, tw.mlp_hidden
    out: List[Tuple[str, Tuple[int, ...], str]] = []
    if tw.has_cls:
        out.append(("cls_token", (1, 1, D), "tok"))
    if tw.n_reg:
        out.append(("reg_token", (1, tw.n_reg, D), "tok"))
    out.append(("pos_embed", (1, tw.n_patches, D), "tok"))
    out.append(("patch_embed.proj.weight", (D, 3, tw.patch, tw.patch), "w"))
    out.append(("patch_embed.proj.bias", (D,), "b"))
    for i in range(tw.depth):
        p = f"blocks.{i}."
        out += [
            (p + "norm1.weight", (D,), "one"), (p + "norm1.bias", (D,), "b"),
            (p + "attn.qkv.weight", (3 * D, D), "w"), (p + "attn.qkv.bias", (3 * D,), "b"),
            (p + "attn.proj.weight", (D, D), "w"), (p + "attn.proj.bias", (D,), "b"),
            (p + "norm2.weight", (D,), "one"), (p + "norm2.bias", (D,), "b"),
            (p + "mlp.fc1.weight", (M, D), "w"), (p + "mlp.fc1.bias", (M,), "b"),
            (p + "mlp.fc2.weight", (D, M), "w"), (p + "mlp.fc2.bias", (D,), "b"),
        ]
        if tw.layerscale:
            out += [(p + "ls1.scale_factor", (D,), "ls"), (p + "ls2.scale_factor", (D,), "ls")]
    return out


def param_shapes(cfg: EmmaXConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Every (key, shape, kind) of the state dict, in a fixed order (the order the synthetic generator consumes RNG)."""
    out: List[Tuple[str, Tuple[int, ...], str]] = []
    for pre, tw in zip(TOWER_PREFIXES, cfg.towers):
        out += [(pre + k, s, kind) for k, s, kind in tower_param_shapes(tw)]
    v, p1, h, _ = cfg.projector_dims
    out += [("projector.fc1.weight", (p1, v), "w"), ("projector.fc1.bias", (p1,), "b"),
            ("projector.fc2.weight", (h, p1), "w"), ("projector.fc2.bias", (h,), "b"),
            ("projector.fc3.weight", (h, h), "w"), ("projector.fc3.bias", (h,), "b")]
    L = cfg.llm
    qd, kvd = L.num_heads * L.head_dim, L.num_kv_heads * L.head_dim
    out.append(("language_model.model.embed_tokens.weight", (L.vocab_size, L.hidden_size), "emb"))
    for i in range(L.num_layers):
        p = f"language_model.model.layers.{i}."
        out += [
            (p + "input_layernorm.weight", (L.hidden_size,), "one"),
            (p + "self_attn.q_proj.weight", (qd, L.hidden_size), "w"),
            (p + "self_attn.k_proj.weight", (kvd, L.hidden_size), "w"),
            (p + "self_attn.v_proj.weight", (kvd, L.hidden_size), "w"),
            (p + "self_attn.o_proj.weight", (L.hidden_size, qd), "wo"),
            (p + "post_attention_layernorm.weight", (L.hidden_size,), "one"),
            (p + "mlp.gate_proj.weight", (L.intermediate_size, L.hidden_size), "w"),
            (p + "mlp.up_proj.weight", (L.intermediate_size, L.hidden_size), "w"),
            (p + "mlp.down_proj.weight", (L.hidden_size, L.intermediate_size), "wo"),
        ]
    out.append(("language_model.model.norm.weight", (L.hidden_size,), "one"))
    out.append(("language_model.lm_head.weight", (L.vocab_size, L.hidden_size), "w"))
    return out


def synthetic_state_dict(cfg: EmmaXConfig, seed: int = 0, device: str = "cpu", dtype: torch.dtype = torch.float32,
                         planted: bool = False, std: float = 0.02, succ_override: Optional[Dict[int, int]] = None) -> Dict[str, torch.Tensor]:
    """Seeded random weights with the real key names / shapes.

    Values: matrices ~ N(0, std); LayerNorm / RMSNorm weights = 1 + 0.1*N; biases 0.01*N; LayerScale 0.1*(1+0.1*N);
    cls/reg/pos tokens ~ N(0, std).  Each tensor gets its own generator seeded from (seed, index) so the values do not
    depend on device or on which other tensors are generated.

    planted=True builds the *margin-boosted* variant used for bit-exact token-id parity: embeddings are unit-scale,
    the residual branches are damped, and `lm_head[succ(t)]` is aligned with `embed[t]` for a fixed successor map
    (`planted_successor`), so greedy decoding has a top-1 margin of many sigma and a known answer.  `succ_override`
    {token: successor} re-routes single entries of that map (tests plant a scripted continuation, e.g. a complete
    "MOVEMENT: .. POLICIES: .." answer, this way).
    """
    sd: Dict[str, torch.Tensor] = {}
    L = cfg.llm
    for idx, (key, shape, kind) in enumerate(param_shapes(cfg)):
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1_000_003 + idx)
        r = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        if kind in ("w", "tok"):
            t = r * std
        elif kind == "wo":
            t = r * (std * (0.25 if planted else 1.0))
        elif kind == "emb":
            t = r * (1.0 if planted else std)
        elif kind == "b":
            t = r * 0.01
        elif kind == "one":
            t = 1.0 + 0.1 * r
        elif kind == "ls":
            t = 0.1 * (1.0 + 0.1 * r)
        else:
            raise ValueError(kind)
        sd[key] = t.to(dtype)
    if planted:
        emb = sd["language_model.model.embed_tokens.weight"].to(torch.float32)
        head = sd["language_model.lm_head.weight"].to(torch.float32)
        succ = planted_successor(cfg)
        for t, nxt in (succ_override or {}).items():
            succ[int(t)] = int(nxt)
        # logits[succ(t)] ~= |embed[t]|^2 * gain / rms  >>  sqrt(hidden)-scale background
        gain = 4.0 * std
        head.index_add_(0, succ.to(emb.device), gain * emb)   # row succ[t] += gain * embed[t]
        sd["language_model.lm_head.weight"] = head.to(dtype)
    return sd


# ---------------------------------------------------------------------------------------------------------------------
# Planted successor map (known-answer greedy decoding for the margin-boosted weights)
# ---------------------------------------------------------------------------------------------------------------------
PLANTED_STRIDE = 7919
PLANTED_PREFIX_TOKEN = 29871      # the `▁` token the reference appends before action tokens (modeling_prismatic.py:513-516)
PLANTED_N_ACTION_TOKENS = 8       # 1 "meaningless" leading token (solver.py:125-126) + 7 DoF


def planted_successor(cfg: EmmaXConfig) -> torch.Tensor:
    """succ[t] for the margin-boosted weights: a fixed map over the vocabulary.

    Ordinary tokens (3..31743) step by a fixed stride inside the ordinary range; token 29871 enters the action range
    (31744..31999); action tokens follow a full-period affine walk, except that the 8th token of the walk that starts
    at the entry token emits EOS (2).  Specials / padding rows map to 29871.
    """
    V = cfg.llm.vocab_size
    av = cfg.action_vocab_size          # 32000
    nb = cfg.n_action_bins
    lo = av - nb                        # 31744
    succ = torch.full((V,), PLANTED_PREFIX_TOKEN, dtype=torch.long)
    t = torch.arange(3, lo)
    succ[3:lo] = 3 + (t - 3 + PLANTED_STRIDE) % (lo - 3)
    a = torch.arange(lo, av)
    succ[lo:av] = lo + ((a - lo) * 5 + 11) % nb
    entry = lo + 37
    succ[PLANTED_PREFIX_TOKEN] = entry
    tok = entry
    for _ in range(PLANTED_N_ACTION_TOKENS - 1):
        tok = int(succ[tok])
    succ[tok] = cfg.eos_token_id
    return succ


def planted_chain(cfg: EmmaXConfig, start: int, n: int) -> List[int]:
    """Known answer: the ids greedy decoding emits after `start` under the planted weights (stops after EOS)."""
    succ = planted_successor(cfg)
    out, t = [], start
    for _ in range(n):
        t = int(succ[t])
        out.append(t)
        if t == cfg.eos_token_id:
            break
    return out


def planted_start_token(cfg: EmmaXConfig, steps_before_prefix: int) -> int:
    """An ordinary token from which the planted walk reaches 29871 after exactly `steps_before_prefix` steps."""
    lo = cfg.action_vocab_size - cfg.n_action_bins
    return 3 + (PLANTED_PREFIX_TOKEN - 3 - steps_before_prefix * PLANTED_STRIDE) % (lo - 3)


# ---------------------------------------------------------------------------------------------------------------------
# Readers
# ---------------------------------------------------------------------------------------------------------------------
def load_hf_state_dict(path: str, device: str = "cpu") -> Dict[str, torch.Tensor]:
    """Read every `*.safetensors` shard under `path` (HF `save_pretrained(max_shard_size="7GB")` layout)."""
    from safetensors import safe_open  # local import: host-side loading only

    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"No *.safetensors shards under {path}")
    sd: Dict[str, torch.Tensor] = {}
    for f in files:
        with safe_open(f, framework="pt", device=device) as sf:
            for k in sf.keys():
                sd[k] = sf.get_tensor(k)
    return sd


PROJECTOR_KEY_MAPPING = {   # native nn.Sequential index -> HF name (convert_openvla_weights_to_hf.py:74-81)
    "projector.0.weight": "projector.fc1.weight", "projector.0.bias": "projector.fc1.bias",
    "projector.2.weight": "projector.fc2.weight", "projector.2.bias": "projector.fc2.bias",
    "projector.4.weight": "projector.fc3.weight", "projector.4.bias": "projector.fc3.bias",
}


def remap_native_state_dict(model_sd: Dict[str, Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """Native `.pt` {"vision_backbone","projector","llm_backbone"} -> HF keys (convert_openvla_weights_to_hf.py:84-116)."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in model_sd["projector"].items():
        out[PROJECTOR_KEY_MAPPING[k]] = v
    for k, v in model_sd["llm_backbone"].items():
        out[k.replace("llm.", "language_model.", 1)] = v
    for k, v in model_sd["vision_backbone"].items():
        if k.startswith("dino_featurizer."):
            k2 = "vision_backbone.featurizer." + k[len("dino_featurizer."):]
        elif k.startswith("siglip_featurizer."):
            k2 = "vision_backbone.fused_featurizer." + k[len("siglip_featurizer."):]
        elif k.startswith("featurizer."):
            k2 = "vision_backbone." + k
        else:
            raise KeyError(f"Unexpected vision key {k}")
        if k2.endswith(".gamma"):   # LayerScale rename (modeling_prismatic.py:52-59)
            k2 = k2[: -len(".gamma")] + ".scale_factor"
        out[k2] = v
    return out


def validate_state_dict(sd: Dict[str, torch.Tensor], cfg: EmmaXConfig) -> None:
    """Raise ValueError listing missing / mis-shaped tensors (extra keys such as `norm.*`, `attn_pool.*` are ignored)."""
    bad = []
    for key, shape, _ in param_shapes(cfg):
        if key not in sd:
            bad.append(f"missing {key}")
        elif tuple(sd[key].shape) != tuple(shape):
            bad.append(f"{key}: expected {tuple(shape)}, got {tuple(sd[key].shape)}")
    if bad:
        raise ValueError("State dict does not match config:\n  " + "\n  ".join(bad[:20]) + (f"\n  ... (+{len(bad) - 20})" if len(bad) > 20 else ""))
