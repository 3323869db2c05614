"""
simpler_env.py -- drop-in for the SimplerEnv policy wrapper `OpenVLAInference`
(experiments/SimplerEnv-OpenVLA/simpler_env/policies/openvla/openvla_model.py:12-145): same constructor arguments, `reset`,
`step(image, task_description)` -> (raw_action, action) with the same keys, gripper post-processing and sticky-gripper state
machine, so `simpler_env/main_inference.py` runs unmodified with the MI355X model behind it.

Host logic only; the 7-DoF action comes from `EmmaXForActionPrediction.predict_action` (the HIP path).  Two dependencies of
the reference file are absent offline and restated here from their published algorithms (pinned in tests):
  * `transforms3d.euler.euler2axangle(roll, pitch, yaw)` (static-frame x-y-z Euler angles -> rotation axis, angle), checked
    against scipy's Rotation.from_euler("xyz").as_rotvec();
  * `cv2.resize(..., interpolation=cv2.INTER_AREA)` (openvla_model.py:147-149): used when importable; otherwise an exact
    area-average resize in numpy (**parity with OpenCV's fixed-point rounding unpinned**; identity at the native 224x224).
Deviation, on purpose: the reference constructor loads the model from the hard-coded hub id "openvla/openvla-7b" whatever
`saved_model_path` says (openvla_model.py:41-47); here `saved_model_path` is what gets loaded, and ready-made `vla` /
`processor` objects may be injected (synthetic weights, tests).
"""

from __future__ import annotations

import math
import os
from typing import Dict, Optional, Sequence, Tuple

import numpy as np


def euler2axangle(ai: float, aj: float, ak: float) -> Tuple[np.ndarray, float]:
    """transforms3d.euler.euler2axangle with the default axes 'sxyz': rotate about the static x, then y, then z axis
    (R = Rz(ak) Ry(aj) Rx(ai)); returns (unit axis [3], angle).  A null rotation gives ([1, 0, 0], 0.0) like transforms3d."""
    ci, si = math.cos(ai / 2.0), math.sin(ai / 2.0)
    cj, sj = math.cos(aj / 2.0), math.sin(aj / 2.0)
    ck, sk = math.cos(ak / 2.0), math.sin(ak / 2.0)
    w = ci * cj * ck + si * sj * sk
    x = si * cj * ck - ci * sj * sk
    y = ci * sj * ck + si * cj * sk
    z = ci * cj * sk - si * sj * ck
    n2 = w * w + x * x + y * y + z * z
    if n2 < 1e-12:
        return np.array([1.0, 0.0, 0.0]), 0.0
    if abs(n2 - 1.0) > 1e-12:
        s = math.sqrt(n2)
        w, x, y, z = w / s, x / s, y / s, z / s
    len2 = x * x + y * y + z * z
    if len2 < 1e-16:   # transforms3d: identity rotation -> arbitrary axis, zero angle
        return np.array([1.0, 0.0, 0.0]), 0.0
    theta = 2.0 * math.acos(max(min(w, 1.0), -1.0))
    return np.array([x, y, z]) / math.sqrt(len2), theta


def resize_area(image: np.ndarray, size: Sequence[int]) -> np.ndarray:
    """uint8 [H,W,3] -> uint8 [size[1], size[0], 3] like cv2.resize(image, tuple(size), interpolation=cv2.INTER_AREA)."""
    ow, oh = int(size[0]), int(size[1])
    h, w = image.shape[:2]
    if (h, w) == (oh, ow):
        return image
    try:
        import cv2  # the reference's own dependency, when present

        return cv2.resize(image, (ow, oh), interpolation=cv2.INTER_AREA)
    except ImportError:
        pass

    def weights(n_in: int, n_out: int) -> np.ndarray:
        # row o of the [n_out, n_in] matrix = fraction of each input cell covered by output cell o, normalised to 1
        scale = n_in / n_out
        m = np.zeros((n_out, n_in), dtype=np.float64)
        for o in range(n_out):
            lo, hi = o * scale, (o + 1) * scale
            i0, i1 = int(math.floor(lo)), min(int(math.ceil(hi)), n_in)
            for i in range(i0, i1):
                m[o, i] = min(hi, i + 1) - max(lo, i)
            m[o] /= m[o].sum()
        return m

    if oh > h or ow > w:
        raise NotImplementedError("INTER_AREA up-scaling (OpenCV falls back to bilinear there) is outside the hot path")
    wy, wx = weights(h, oh), weights(w, ow)
    x = np.einsum("oh,hwc->owc", wy, image.astype(np.float64))
    x = np.einsum("pw,owc->opc", wx, x)
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


class OpenVLAInference:
    def __init__(self, saved_model_path: str = "openvla/openvla-7b", unnorm_key: Optional[str] = None,
                 policy_setup: str = "widowx_bridge", horizon: int = 1, pred_action_horizon: int = 1, exec_horizon: int = 1,
                 image_size: Sequence[int] = (224, 224), action_scale: float = 1.0, vla=None, processor=None,
                 device: str = "cuda:0") -> None:
        os.environ["TOKENIZERS_PARALLELISM"] = "false"
        if policy_setup == "widowx_bridge":
            unnorm_key = "bridge_orig" if unnorm_key is None else unnorm_key
            self.sticky_gripper_num_repeat = 1
        elif policy_setup == "google_robot":
            unnorm_key = "fractal20220817_data" if unnorm_key is None else unnorm_key
            self.sticky_gripper_num_repeat = 15
        else:
            raise NotImplementedError(
                f"Policy setup {policy_setup} not supported for octo models. The other datasets can be found in the huggingface config.json file."
            )
        self.policy_setup = policy_setup
        self.unnorm_key = unnorm_key
        self.device = device
        if processor is None or vla is None:
            from .modeling import EmmaXForActionPrediction
            from .processing import EmmaXProcessor

            processor = processor if processor is not None else EmmaXProcessor.from_pretrained(saved_model_path)
            if vla is None:
                import torch

                vla = EmmaXForActionPrediction.from_pretrained(saved_model_path, torch_dtype=torch.bfloat16, low_cpu_mem_usage=True,
                                                               trust_remote_code=True).to(device)
        self.processor, self.vla = processor, vla

        self.image_size = list(image_size)
        self.action_scale = action_scale
        self.horizon = horizon
        self.pred_action_horizon = pred_action_horizon
        self.exec_horizon = exec_horizon

        self.sticky_action_is_on = False
        self.gripper_action_repeat = 0
        self.sticky_gripper_action = 0.0
        self.previous_gripper_action = None

        self.task = None
        self.task_description = None
        self.num_image_history = 0

    def reset(self, task_description: str) -> None:
        self.task_description = task_description
        self.num_image_history = 0

        self.sticky_action_is_on = False
        self.gripper_action_repeat = 0
        self.sticky_gripper_action = 0.0
        self.previous_gripper_action = None

    def _predict(self, prompt: Optional[str], image: np.ndarray) -> np.ndarray:
        import torch

        inputs = self.processor(prompt, image).to(self.device, dtype=torch.bfloat16)
        return self.vla.predict_action(**inputs, unnorm_key=self.unnorm_key, do_sample=False)

    def step(self, image: np.ndarray, task_description: Optional[str] = None, *args, **kwargs) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
        """image uint8 [H,W,3] -> (raw_action {world_vector[3], rotation_delta[3], open_gripper[1]},
        action {world_vector[3], rot_axangle[3], gripper[1], terminate_episode[1]}) -- openvla_model.py:72-145."""
        if task_description is not None:
            if task_description != self.task_description:
                self.reset(task_description)

        assert image.dtype == np.uint8
        image = self._resize_image(image)
        prompt = task_description   # the reference passes the bare description (None when omitted) as the prompt

        raw_actions = np.asarray(self._predict(prompt, image))[None]
        raw_action = {
            "world_vector": np.array(raw_actions[0, :3]),
            "rotation_delta": np.array(raw_actions[0, 3:6]),
            "open_gripper": np.array(raw_actions[0, 6:7]),  # range [0, 1]; 1 = open; 0 = close
        }

        action = {}
        action["world_vector"] = raw_action["world_vector"] * self.action_scale
        roll, pitch, yaw = np.asarray(raw_action["rotation_delta"], dtype=np.float64)
        ax, angle = euler2axangle(roll, pitch, yaw)
        action["rot_axangle"] = ax * angle * self.action_scale

        if self.policy_setup == "google_robot":
            current_gripper_action = raw_action["open_gripper"]
            if self.previous_gripper_action is None:
                relative_gripper_action = np.array([0])
            else:
                relative_gripper_action = self.previous_gripper_action - current_gripper_action
            self.previous_gripper_action = current_gripper_action

            if np.abs(relative_gripper_action) > 0.5 and (not self.sticky_action_is_on):
                self.sticky_action_is_on = True
                self.sticky_gripper_action = relative_gripper_action

            if self.sticky_action_is_on:
                self.gripper_action_repeat += 1
                relative_gripper_action = self.sticky_gripper_action

            if self.gripper_action_repeat == self.sticky_gripper_num_repeat:
                self.sticky_action_is_on = False
                self.gripper_action_repeat = 0
                self.sticky_gripper_action = 0.0

            action["gripper"] = relative_gripper_action
        elif self.policy_setup == "widowx_bridge":
            action["gripper"] = 2.0 * (raw_action["open_gripper"] > 0.5) - 1.0

        action["terminate_episode"] = np.array([0.0])
        return raw_action, action

    def _resize_image(self, image: np.ndarray) -> np.ndarray:
        return resize_area(image, tuple(self.image_size))
