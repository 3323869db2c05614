"""Write an HF-format Emma-X checkpoint directory with synthetic weights: config.json (OpenVLAConfig fields),
model-0000X-of-0000N.safetensors shards with the converter's key names, dataset_statistics.json.  No real checkpoint or
LLaMA tokenizer exists offline; this exercises the `from_pretrained` ingest path end to end (the stub tokenizer is used
when the directory has no tokenizer files)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "emma-x_amd")]
import torch
from safetensors.torch import save_file

from emmax.config import EmmaXConfig, default_norm_stats
from emmax.weights import synthetic_state_dict


def write_checkpoint(out: str, cfg: EmmaXConfig, seed: int = 0, planted: bool = True, shards: int = 2, tiny_towers: bool = False,
                     auto_map: bool = False) -> None:
    os.makedirs(out, exist_ok=True)
    if not cfg.norm_stats:   # a synthetic checkpoint ships synthetic (made-up) statistics, like from_synthetic; never an empty dict
        cfg.norm_stats = default_norm_stats()
    sd = {k: v.to(torch.bfloat16).contiguous() for k, v in synthetic_state_dict(cfg, seed=seed, planted=planted).items()}
    keys = sorted(sd)
    per = (len(keys) + shards - 1) // shards
    index = {"metadata": {"total_size": sum(v.numel() * 2 for v in sd.values())}, "weight_map": {}}
    for i in range(shards):
        name = f"model-{i + 1:05d}-of-{shards:05d}.safetensors"
        part = {k: sd[k] for k in keys[i * per:(i + 1) * per]}
        save_file(part, os.path.join(out, name), metadata={"format": "pt"})
        index["weight_map"].update({k: name for k in part})
    with open(os.path.join(out, "model.safetensors.index.json"), "w") as f:
        json.dump(index, f)
    L = cfg.llm
    conf = {
        "model_type": "openvla", "architectures": ["OpenVLAForActionPrediction"],
        "vision_backbone_id": cfg.vision_backbone_id, "llm_backbone_id": cfg.llm_backbone_id,
        "arch_specifier": cfg.arch_specifier, "image_resize_strategy": cfg.image_resize_strategy,
        "use_fused_vision_backbone": True, "image_sizes": [224, 224],
        "timm_model_ids": [t.timm_id for t in cfg.towers], "llm_max_length": L.max_position,
        "n_action_bins": cfg.n_action_bins, "pad_token_id": cfg.pad_token_id, "pad_to_multiple_of": cfg.pad_to_multiple_of,
        "text_config": {"model_type": "llama", "hidden_size": L.hidden_size, "intermediate_size": L.intermediate_size,
                        "num_hidden_layers": L.num_layers, "num_attention_heads": L.num_heads,
                        "num_key_value_heads": L.num_kv_heads, "head_dim": L.head_dim, "vocab_size": L.vocab_size,
                        "rms_norm_eps": L.rms_eps, "rope_theta": L.rope_theta, "pad_token_id": cfg.pad_token_id},
        "norm_stats": cfg.norm_stats, "torch_dtype": "bfloat16",
    }
    if auto_map:   # what real OpenVLA / Emma-X checkpoints carry next to their bundled modeling code
        conf["auto_map"] = {"AutoConfig": "configuration_prismatic.OpenVLAConfig",
                            "AutoModelForVision2Seq": "modeling_prismatic.OpenVLAForActionPrediction"}
    if tiny_towers:   # non-standard: the real config carries no tower dims (they come from the timm ids)
        conf["emmax_tower_overrides"] = [{"embed_dim": t.embed_dim, "depth": t.depth, "num_heads": t.num_heads,
                                          "mlp_hidden": t.mlp_hidden} for t in cfg.towers]
    with open(os.path.join(out, "config.json"), "w") as f:
        json.dump(conf, f, indent=1)
    with open(os.path.join(out, "dataset_statistics.json"), "w") as f:
        json.dump(cfg.norm_stats, f)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--full", action="store_true", help="7B shapes (15 GB)")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    write_checkpoint(a.out, EmmaXConfig.emma_x_7b() if a.full else EmmaXConfig.tiny(), a.seed, tiny_towers=not a.full)
    print("wrote", a.out)
