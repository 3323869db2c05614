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


# Per-embodiment constants of the SimplerEnv wrapper: the statistics key the model un-normalises with by default and the
# number of control steps a google_robot gripper command is held for (openvla_model.py:24-34).
_EMBODIMENTS = {
    "widowx_bridge": {"unnorm_key": "bridge_orig", "hold_steps": 1, "gripper": "absolute"},
    "google_robot": {"unnorm_key": "fractal20220817_data", "hold_steps": 15, "gripper": "sticky_delta"},
}


class StickyGripper:
    """google_robot gripper post-processing (behaviour of openvla_model.py:104-131, pinned by tests/golden/simpler_env.json).

    The model emits an absolute opening in [0, 1] (1 = open); the simulator wants the CHANGE of opening, and wants a large
    change repeated for `hold_steps` consecutive control steps.  State: the previous opening, and -- while a command is being
    held -- the held delta and how many steps it has been emitted for."""

    def __init__(self, hold_steps: int) -> None:
        self.hold_steps = int(hold_steps)
        self.clear()

    def clear(self) -> None:
        self.last_opening: Optional[np.ndarray] = None
        self.held: Optional[np.ndarray] = None   # delta being repeated, None when idle
        self.emitted = 0

    def __call__(self, opening: np.ndarray) -> np.ndarray:
        delta = np.array([0]) if self.last_opening is None else self.last_opening - opening
        self.last_opening = opening
        if self.held is None and np.abs(delta) > 0.5:
            self.held = delta
        if self.held is not None:
            delta = self.held
            self.emitted += 1
        if self.emitted == self.hold_steps:   # also true for an idle gripper when hold_steps == 0; harmless
            self.held, self.emitted = None, 0
        return delta


class OpenVLAInference:
    """Same constructor arguments, `reset`, `step` and result keys as the SimplerEnv wrapper (openvla_model.py:12-145)."""

    ACTION_LABELS = ("x", "y", "z", "roll", "pitch", "yaw", "grasp")

    def __init__(self, saved_model_path: str = "openvla/openvla-7b", unnorm_key: Optional[str] = None,
                 policy_setup: str = "widowx_bridge", horizon: int = 1, pred_action_horizon: int = 1, exec_horizon: int = 1,
                 image_size: Sequence[int] = (224, 224), action_scale: float = 1.0, vla=None, processor=None,
                 device: str = "cuda:0") -> None:
        os.environ["TOKENIZERS_PARALLELISM"] = "false"
        setup = _EMBODIMENTS.get(policy_setup)
        if setup is None:
            raise NotImplementedError(f"Policy setup {policy_setup!r} is not one of {sorted(_EMBODIMENTS)}; other embodiments "
                                      "need their statistics key from the checkpoint's config.json")
        self.policy_setup = policy_setup
        self.unnorm_key = setup["unnorm_key"] if unnorm_key is None else unnorm_key
        self.sticky_gripper_num_repeat = setup["hold_steps"]
        self._gripper = StickyGripper(setup["hold_steps"]) if setup["gripper"] == "sticky_delta" else None
        self.device = device
        self.image_size = [int(v) for v in image_size]
        self.action_scale = action_scale
        self.horizon, self.pred_action_horizon, self.exec_horizon = horizon, pred_action_horizon, exec_horizon
        self.task = None
        self.task_description: Optional[str] = None
        self.num_image_history = 0
        self.processor, self.vla = self._load(saved_model_path, processor, vla, device)

    @staticmethod
    def _load(path, processor, vla, device):
        if processor is None:
            from .processing import EmmaXProcessor

            processor = EmmaXProcessor.from_pretrained(path)
        if vla is None:
            import torch

            from .modeling import EmmaXForActionPrediction

            vla = EmmaXForActionPrediction.from_pretrained(path, torch_dtype=torch.bfloat16, low_cpu_mem_usage=True,
                                                           trust_remote_code=True).to(device)
        return processor, vla

    def reset(self, task_description: str) -> None:
        self.task_description = task_description
        self.num_image_history = 0
        if self._gripper is not None:
            self._gripper.clear()

    def _predict(self, prompt: Optional[str], image: np.ndarray) -> np.ndarray:
        import torch

        inputs = self.processor(prompt, image).to(self.device, dtype=torch.bfloat16)
        return self.vla.predict_action(**inputs, unnorm_key=self.unnorm_key, do_sample=False)

    def step(self, image: np.ndarray, task_description: Optional[str] = None, *args, **kwargs) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
        """image uint8 [H,W,3] -> (raw_action {world_vector[3], rotation_delta[3], open_gripper[1]},
        action {world_vector[3], rot_axangle[3], gripper[1], terminate_episode[1]}) -- contract of openvla_model.py:72-145.
        A task description different from the current one starts a new episode; the bare description (None when omitted) is
        the prompt, as in the reference."""
        if task_description is not None and task_description != self.task_description:
            self.reset(task_description)
        if image.dtype != np.uint8:
            raise AssertionError("step() takes a uint8 camera frame")
        vec = np.asarray(self._predict(task_description, self._resize_image(image))).reshape(-1)
        xyz, rpy, opening = np.array(vec[0:3]), np.array(vec[3:6]), np.array(vec[6:7])
        raw_action = {"world_vector": xyz, "rotation_delta": rpy, "open_gripper": opening}

        axis, angle = euler2axangle(*(float(v) for v in rpy))
        if self._gripper is not None:
            grip = self._gripper(opening)                    # change of opening, held over several steps
        else:
            grip = np.where(opening > 0.5, 1.0, -1.0)       # widowx: +1 open / -1 close
        action = {
            "world_vector": xyz * self.action_scale,
            "rot_axangle": axis * angle * self.action_scale,
            "gripper": grip,
            "terminate_episode": np.array([0.0]),
        }
        return raw_action, action

    def _resize_image(self, image: np.ndarray) -> np.ndarray:
        return resize_area(image, tuple(self.image_size))

    def visualize_epoch(self, predicted_raw_actions: Sequence[Dict[str, np.ndarray]], images: Sequence[np.ndarray], save_path: str) -> None:
        """Episode summary the SimplerEnv evaluator asks for after a roll-out (maniskill2_evaluator.py:170): one trace per
        action dimension over every third resized frame.  A figure when matplotlib is importable, else the traces as .npy."""
        traces = np.stack([np.concatenate([np.ravel(a[k]) for k in ("world_vector", "rotation_delta", "open_gripper")])
                           for a in predicted_raw_actions])
        try:
            import matplotlib

            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except ImportError:
            np.save(os.path.splitext(save_path)[0] + "_actions.npy", traces)
            return
        strip = np.concatenate([self._resize_image(im) for im in images[::3]], axis=1)
        fig = plt.figure(figsize=(45, 10))
        grid = fig.add_gridspec(2, len(self.ACTION_LABELS))
        top = fig.add_subplot(grid[0, :])
        top.imshow(strip)
        top.set_xlabel("episode frames (every third)")
        for d, name in enumerate(self.ACTION_LABELS):
            ax = fig.add_subplot(grid[1, d])
            ax.plot(traces[:, d])
            ax.set_title(name)
            ax.set_xlabel("step")
        fig.savefig(save_path)
        plt.close(fig)
