"""
config.py -- model constants for the Emma-X hot path (host side, no compute).

Mirrors the config contract of the reference's HF classes:
  * `PrismaticConfig` / `OpenVLAConfig`           prismatic/extern/hf/configuration_prismatic.py:72-140
  * backbone tables (timm ids, image sizes)        configuration_prismatic.py:15-69
  * Emma-X-7B = `prism-dinosiglip-224px+7b`        prismatic/conf/models.py:491-497
  * LLaMA-2 pad-token / vocab padding to 32064     prismatic/models/backbones/llm/llama2.py:73-76
Values tagged [not in ref] come from timm 0.9.10 model defs / Llama-2-7b-hf config.json.
"""

from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple


@dataclass
class TowerConfig:
    """One timm ViT tower as the reference instantiates it (modeling_prismatic.py:78-101)."""

    timm_id: str
    embed_dim: int
    depth: int
    num_heads: int
    mlp_hidden: int
    n_reg: int = 0            # register tokens (DINOv2 reg4)
    has_cls: bool = False     # class token present
    layerscale: bool = False  # LayerScale (`ls{1,2}.scale_factor`, renamed from `.gamma`: modeling_prismatic.py:52-59)
    mean: Tuple[float, float, float] = (0.5, 0.5, 0.5)
    std: Tuple[float, float, float] = (0.5, 0.5, 0.5)
    patch: int = 14
    image_size: int = 224
    ln_eps: float = 1e-6

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def n_prefix(self) -> int:
        return (1 if self.has_cls else 0) + self.n_reg

    @property
    def n_patches(self) -> int:
        return (self.image_size // self.patch) ** 2

    @property
    def n_tokens(self) -> int:
        return self.n_prefix + self.n_patches

    @property
    def take_index(self) -> int:
        # get_intermediate_layers(n={len(blocks) - 2})  (modeling_prismatic.py:86,100)
        return self.depth - 2


@dataclass
class LlmConfig:
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_layers: int = 32
    num_heads: int = 32
    num_kv_heads: int = 32
    head_dim: int = 128
    vocab_size: int = 32064   # 32000 + <PAD>, padded to a multiple of 64 (llama2.py:73-76)
    rms_eps: float = 1e-5     # Llama-2 checkpoint value; always read text_config.rms_norm_eps
    rope_theta: float = 10000.0
    max_position: int = 2048  # llm_max_length (configuration_prismatic.py:84)


@dataclass
class EmmaXConfig:
    towers: List[TowerConfig]
    llm: LlmConfig
    n_action_bins: int = 256
    pad_to_multiple_of: int = 64
    bos_token_id: int = 1
    eos_token_id: int = 2
    pad_token_id: int = 32000
    norm_stats: Dict[str, Any] = field(default_factory=dict)
    vision_backbone_id: str = "dinosiglip-vit-so-224px"
    llm_backbone_id: str = "llama2-7b-pure"
    arch_specifier: str = "no-align+fused-gelu-mlp"
    image_resize_strategy: str = "resize-naive"
    # MI355X extension (BASELINE config 5): "fp8" streams an e4m3 per-row-scaled copy of the LLM projections in decode
    decode_weight_dtype: str = "bf16"

    # --- derived ---
    @property
    def vision_dim(self) -> int:
        return sum(t.embed_dim for t in self.towers)

    @property
    def projector_dims(self) -> Tuple[int, int, int, int]:
        # FusedMLPProjector: vision_dim -> 4*vision_dim -> llm_dim -> llm_dim  (nn_utils.py:37-53)
        v, h = self.vision_dim, self.llm.hidden_size
        return (v, 4 * v, h, h)

    @property
    def n_patches(self) -> int:
        return self.towers[0].n_patches

    @property
    def action_vocab_size(self) -> int:
        # vocab size used for de-tokenisation: text_config.vocab_size - pad_to_multiple_of (modeling_prismatic.py:504)
        return self.llm.vocab_size - self.pad_to_multiple_of

    # --- factories ---
    @staticmethod
    def emma_x_7b(norm_stats: Optional[Dict[str, Any]] = None) -> "EmmaXConfig":
        dino = TowerConfig("vit_large_patch14_reg4_dinov2.lvd142m", 1024, 24, 16, 4096, n_reg=4, has_cls=True,
                           layerscale=True, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225))
        siglip = TowerConfig("vit_so400m_patch14_siglip_224", 1152, 27, 16, 4304)
        # norm_stats=None -> EMPTY: `_check_unnorm_key` then raises instead of un-normalising with invented statistics; only the
        # synthetic factories (`from_synthetic`, `tiny`) install `default_norm_stats()`
        return EmmaXConfig([dino, siglip], LlmConfig(), norm_stats=norm_stats if norm_stats is not None else {})

    @staticmethod
    def tiny(gqa: bool = False, norm_stats: Optional[Dict[str, Any]] = None) -> "EmmaXConfig":
        """Small config that exercises every kernel template of the 7B model (head dims 64 / 72 / 128, a ragged MLP
        width, reg tokens + LayerScale, vocab 32064) at a size the CPU oracle finishes in seconds."""
        dino = TowerConfig("tiny_dinov2_reg4", 128, 4, 2, 256, n_reg=4, has_cls=True, layerscale=True,
                           mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225))
        siglip = TowerConfig("tiny_siglip", 144, 4, 2, 304)
        llm = LlmConfig(hidden_size=256, intermediate_size=688, num_layers=3, num_heads=4 if gqa else 2,
                        num_kv_heads=2, head_dim=128, vocab_size=32064, max_position=2048)
        return EmmaXConfig([dino, siglip], llm, norm_stats=norm_stats or default_norm_stats())

    # --- HF config.json contract (configuration_prismatic.py:72-140) ---
    @staticmethod
    def from_hf_dict(d: Dict[str, Any]) -> "EmmaXConfig":
        vb = d.get("vision_backbone_id", "dinosiglip-vit-so-224px")
        if vb != "dinosiglip-vit-so-224px":
            raise ValueError(f"Vision backbone `{vb}` is outside the Emma-X-7B hot path (only dinosiglip-vit-so-224px)")
        llm_id = d.get("llm_backbone_id", "llama2-7b-pure")
        if llm_id not in ("llama2-7b-pure",):
            raise ValueError(f"LLM backbone `{llm_id}` is outside the Emma-X-7B hot path (only llama2-7b-pure)")
        cfg = EmmaXConfig.emma_x_7b(norm_stats=d.get("norm_stats") or {})
        tc = d.get("text_config") or {}
        L = cfg.llm
        L.hidden_size = tc.get("hidden_size", L.hidden_size)
        L.intermediate_size = tc.get("intermediate_size", L.intermediate_size)
        L.num_layers = tc.get("num_hidden_layers", L.num_layers)
        L.num_heads = tc.get("num_attention_heads", L.num_heads)
        L.num_kv_heads = tc.get("num_key_value_heads", L.num_heads)
        L.head_dim = tc.get("head_dim", L.hidden_size // L.num_heads)
        L.vocab_size = tc.get("vocab_size", L.vocab_size)
        L.rms_eps = tc.get("rms_norm_eps", L.rms_eps)
        L.rope_theta = tc.get("rope_theta", L.rope_theta)
        L.max_position = d.get("llm_max_length", L.max_position)
        for tw, ov in zip(cfg.towers, d.get("emmax_tower_overrides") or []):   # synthetic tiny checkpoints only
            for k in ("embed_dim", "depth", "num_heads", "mlp_hidden"):
                if k in ov:
                    setattr(tw, k, int(ov[k]))
        cfg.n_action_bins = d.get("n_action_bins", 256)
        cfg.pad_to_multiple_of = d.get("pad_to_multiple_of", 64)
        cfg.pad_token_id = d.get("pad_token_id", 32000)
        cfg.arch_specifier = d.get("arch_specifier", cfg.arch_specifier)
        cfg.image_resize_strategy = d.get("image_resize_strategy", cfg.image_resize_strategy)
        return cfg

    @staticmethod
    def from_native_dict(d: Dict[str, Any]) -> "EmmaXConfig":
        """`config.json` of a native Prismatic run directory (prismatic/models/load.py:176-179: {"vla": {"base_vlm": ...}}): the
        Emma-X hot path is the `prism-dinosiglip-224px+7b` family only (prismatic/conf/models.py:491-497)."""
        base = (d.get("vla") or {}).get("base_vlm") or (d.get("model") or {}).get("model_id") or ""
        if base and "dinosiglip-224px" not in base:
            raise ValueError(f"base VLM `{base}` is outside the Emma-X-7B hot path (only prism-dinosiglip-224px+7b)")
        return EmmaXConfig.emma_x_7b(norm_stats={})

    @staticmethod
    def from_pretrained(path: str) -> "EmmaXConfig":
        with open(os.path.join(path, "config.json")) as f:
            cfg = EmmaXConfig.from_hf_dict(json.load(f))
        ds = os.path.join(path, "dataset_statistics.json")   # experiments/robot/openvla_utils.py:60-64
        if os.path.isfile(ds):
            with open(ds) as f:
                cfg.norm_stats = json.load(f)
        return cfg


def default_norm_stats() -> Dict[str, Any]:
    """Synthetic `bridge_orig`-shaped statistics (values are made up; the real ones ship with a checkpoint)."""
    return {
        "bridge_orig": {
            "action": {
                "q01": [-0.03, -0.04, -0.035, -0.08, -0.09, -0.2, 0.0],
                "q99": [0.028, 0.041, 0.04, 0.081, 0.078, 0.2, 1.0],
                "mask": [True, True, True, True, True, True, False],
            },
            "proprio": {
                "Q1": [-0.1, -0.2, -0.1, -0.3, -0.3, -0.3, 0.0],
                "Q99": [0.4, 0.3, 0.35, 0.3, 0.3, 0.3, 1.0],
                "mask": [True, True, True, True, True, True, False],
            },
        }
    }
