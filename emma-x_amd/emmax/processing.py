"""
processing.py -- `AutoProcessor`-compatible host-side preprocessing.

Behavioural mirror of `PrismaticImageProcessor` / `PrismaticProcessor`
(prismatic/extern/hf/processing_prismatic.py:32-252) for the Emma-X configuration (`resize-naive`, two towers):
per tower  resize(bicubic) -> center-crop -> to_tensor -> normalize, channel-stacked to [B,6,224,224] float32, plus
tokenisation of the prompt.  torchvision/timm are not needed: torchvision's PIL resize IS `PIL.Image.resize`.

Beyond the reference surface the BatchFeature also carries `frames_u8` ([B,224,224,3] uint8) when every input frame is
already 224x224, so the device path can fuse the normalisation into the patch gather and skip the fp32 pixel tensor.
"""

from __future__ import annotations

import json
import os
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .config import EmmaXConfig
from .prompting import bridge_task_label, build_prompt
from .tokenizer_stub import StubTokenizer

try:  # PIL is only needed for non-224 inputs and PIL.Image inputs
    from PIL import Image
except Exception:  # pragma: no cover
    Image = None


class BatchFeature(dict):
    """dict with attribute access and the `.to(device, dtype=...)` the callers use (openvla_utils.py:166)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to(self, device=None, dtype=None):
        out = BatchFeature()
        for k, v in self.items():
            if isinstance(v, torch.Tensor):
                if v.is_floating_point() and dtype is not None:
                    out[k] = v.to(device=device, dtype=dtype)
                else:
                    out[k] = v.to(device=device)
                if hasattr(v, "_emmax_frames_tag"):   # a device / dtype move keeps the (pixel_values, frames_u8) pairing
                    out[k]._emmax_frames_tag = v._emmax_frames_tag
            else:
                out[k] = v
        return out


def _to_uint8_hwc(img) -> np.ndarray:
    if Image is not None and isinstance(img, Image.Image):
        return np.asarray(img.convert("RGB"), dtype=np.uint8)
    a = np.asarray(img)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("images must be PIL.Image or uint8 arrays of shape [H,W,3]")
    return a


def letterbox_pad(img_u8: np.ndarray, fill: Sequence[int]) -> np.ndarray:
    """Pad to square with `fill` (processing_prismatic.py:23-29)."""
    h, w = img_u8.shape[:2]
    m = max(h, w)
    top, left = (m - h) // 2, (m - w) // 2
    out = np.empty((m, m, 3), dtype=np.uint8)
    out[...] = np.asarray(fill, dtype=np.uint8)
    out[top:top + h, left:left + w] = img_u8
    return out


class EmmaXImageProcessor:
    model_input_names = ["pixel_values"]

    def __init__(self, cfg: EmmaXConfig) -> None:
        self.cfg = cfg
        self.image_resize_strategy = cfg.image_resize_strategy
        self.input_sizes = [(3, t.image_size, t.image_size) for t in cfg.towers]
        self.means = [t.mean for t in cfg.towers]
        self.stds = [t.std for t in cfg.towers]
        self.use_fused_vision_backbone = len(cfg.towers) == 2

    def _resize(self, a: np.ndarray, size: int) -> np.ndarray:
        if a.shape[0] == size and a.shape[1] == size:
            return a
        if Image is None:
            raise RuntimeError("PIL is required to resize non-native frames")
        if self.image_resize_strategy == "letterbox":
            a = letterbox_pad(a, tuple(int(x * 255) for x in self.means[0]))
        # torchvision TVF.resize on a PIL image == PIL resize (antialiased bicubic); resize-naive squashes to a square
        return np.asarray(Image.fromarray(a).resize((size, size), Image.BICUBIC), dtype=np.uint8)

    def apply_transform(self, img) -> Tuple[torch.Tensor, np.ndarray]:
        a = _to_uint8_hwc(img)
        size = self.cfg.towers[0].image_size
        a = self._resize(a, size)
        x = torch.from_numpy(np.array(a, dtype=np.uint8, copy=True)).permute(2, 0, 1).to(torch.float32) / 255.0
        per_tower = []
        for mean, std in zip(self.means, self.stds):
            m = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
            s = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
            per_tower.append((x - m) / s)
        return torch.vstack(per_tower), a

    def preprocess(self, images, return_tensors: Optional[str] = "pt", **_) -> BatchFeature:
        if not isinstance(images, (list, tuple)):
            images = [images]
        pix, raw = zip(*(self.apply_transform(im) for im in images))
        pv, fr = torch.stack(pix), torch.from_numpy(np.stack(raw))
        # `pixel_values` is the model input; `frames_u8` is the same frame before normalisation.  The shared tag lets the model
        # take the fused uint8 route ONLY while `pixel_values` is still this very tensor (modeling._encode_images)
        pv._emmax_frames_tag = fr._emmax_frames_tag = object()
        return BatchFeature(pixel_values=pv, frames_u8=fr)

    __call__ = preprocess


class EmmaXProcessor:
    """`processor(text, images)` -> BatchFeature{input_ids, attention_mask, pixel_values[, frames_u8]}."""

    attributes = ["image_processor", "tokenizer"]
    model_input_names = ["input_ids", "attention_mask", "pixel_values"]

    def __init__(self, image_processor: EmmaXImageProcessor, tokenizer=None) -> None:
        self.image_processor = image_processor
        if tokenizer is None:
            raise ValueError("EmmaXProcessor needs a tokenizer (from_pretrained(dir) loads the checkpoint's; from_synthetic() names the stub)")
        self.tokenizer = tokenizer

    @classmethod
    def from_pretrained(cls, path: Optional[str] = None, cfg: Optional[EmmaXConfig] = None, **_) -> "EmmaXProcessor":
        """Reads config.json and the tokenizer files from a checkpoint directory (processing_prismatic.py:216 surface).

        A directory WITHOUT tokenizer files is an error: tokenising prompts with a stand-in would feed the model garbage ids
        without any sign of it.  With no path at all this is `from_synthetic` (the stub tokenizer, named as such)."""
        if not path:
            return cls.from_synthetic(cfg)
        if cfg is None:
            cfg = EmmaXConfig.from_pretrained(path)
        if not any(os.path.isfile(os.path.join(path, f)) for f in ("tokenizer.json", "tokenizer.model")):
            raise FileNotFoundError(
                f"{path!r} holds no tokenizer.json / tokenizer.model: refusing to tokenise prompts with a stand-in. Copy the "
                "checkpoint's LLaMA tokenizer files there, or build the processor explicitly with "
                "EmmaXProcessor.from_synthetic(cfg) for synthetic-weight runs.")
        from transformers import AutoTokenizer  # host-side only, never on the device path

        tok = AutoTokenizer.from_pretrained(path, model_max_length=cfg.llm.max_position, padding_side="right")
        return cls(EmmaXImageProcessor(cfg), tok)

    @classmethod
    def from_synthetic(cls, cfg: Optional[EmmaXConfig] = None) -> "EmmaXProcessor":
        """Processor for synthetic-weight runs (tests, bench): the deterministic character-level `StubTokenizer`; text<->id
        parity with the LLaMA tokenizer is unpinned there."""
        return cls(EmmaXImageProcessor(cfg if cfg is not None else EmmaXConfig.emma_x_7b()), StubTokenizer())

    def __call__(self, text: Union[str, List[str]], images, padding: bool = False, truncation: Optional[bool] = None,
                 max_length: Optional[int] = None, return_tensors: str = "pt") -> BatchFeature:
        feat = self.image_processor(images, return_tensors=return_tensors)
        enc = self.tokenizer(text, return_tensors=return_tensors, padding=padding, truncation=bool(truncation),
                             max_length=max_length)
        if feat["pixel_values"].shape[0] != enc["input_ids"].shape[0]:
            raise ValueError("Batch is malformed; expected same number of images and text inputs!")
        return BatchFeature(input_ids=enc["input_ids"], attention_mask=enc["attention_mask"], **feat)

    def get_prompt(self, task_label: str, image, gripper_xy: Optional[Sequence[int]] = None):
        """README.md:44 `prompt, image = processor.get_prompt(task_label, image)` (hub-only in the reference; rebuilt from
        the Bridge template run_bridgev2_eval.py:167-168 + PurePromptBuilder; no gripper detector: parity unpinned)."""
        return build_prompt(bridge_task_label(task_label, gripper_xy)), image

    def decode(self, *a, **k):
        return self.tokenizer.decode(*a, **k)

    def batch_decode(self, *a, **k):
        return self.tokenizer.batch_decode(*a, **k)
