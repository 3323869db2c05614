"""
callers.py -- drop-in replacements for the policy glue that sits directly above the hot path.

Behavioural mirror of `get_vla_action` / `get_seq_action` (experiments/robot/openvla_utils.py:127-218): same arguments,
same prompt construction, same return values.  `center_crop=True` (a TensorFlow crop-and-resize augmentation, disabled
for Bridge: experiments/robot/bridge/run_bridgev2_eval.py:102) is outside the hot path and raises.
"""

from __future__ import annotations

import numpy as np
import torch

try:
    from PIL import Image
except Exception:  # pragma: no cover
    Image = None

OPENVLA_V01_SYSTEM_PROMPT = (
    "A chat between a curious user and an artificial intelligence assistant. "
    "The assistant gives helpful, detailed, and polite answers to the user's questions."
)


def _frame(obs):
    img = obs["full_image"]
    if Image is not None:
        return Image.fromarray(np.asarray(img)).convert("RGB")
    return np.asarray(img)


def get_vla_action(vla, processor, base_vla_name, obs, task_label, unnorm_key, center_crop=False):
    """One 7-DoF action through `predict_action` (OpenVLA-style prompt, 7 new tokens)."""
    if center_crop:
        raise NotImplementedError("center_crop (TensorFlow crop_and_resize) is outside the MI355X hot path")
    image = _frame(obs)
    if "openvla-v01" in base_vla_name:
        prompt = f"{OPENVLA_V01_SYSTEM_PROMPT} USER: What action should the robot take to {task_label.lower()}? ASSISTANT:"
    else:
        prompt = f"In: What action should the robot take to {task_label.lower()}?\nOut:"
    inputs = processor(prompt, image).to(vla.device, dtype=torch.bfloat16)
    return vla.predict_action(**inputs, unnorm_key=unnorm_key, do_sample=False)


def get_seq_action(vla, processor, base_vla_name, obs, task_label, unnorm_key, type, center_crop=False):
    """Emma-X grounded chain-of-thought rollout: (list of 7-DoF policies | proprio target, generated text)."""
    if center_crop:
        raise NotImplementedError("center_crop (TensorFlow crop_and_resize) is outside the MI355X hot path")
    image = _frame(obs)
    builder = vla.get_prompt_builder()
    builder.add_turn(role="human", message=task_label)
    prompt = builder.get_prompt()
    if getattr(vla, "tokenizer", None) is None:   # a checkpoint directory without tokenizer files: share the processor's
        vla.tokenizer = processor.tokenizer
    # the reference's call, verbatim (experiments/robot/openvla_utils.py:215-217)
    return vla.generate_actions(
        image=image, prompt_text=prompt, type=type, temperature=0.0, max_new_tokens=512, min_length=1, do_sample=False
    )
