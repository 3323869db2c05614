"""emmax -- MI355X-native Emma-X VLA hot path (host-side mirror of prismatic/extern/hf + prismatic/models/vlms)."""

from .config import EmmaXConfig, LlmConfig, TowerConfig  # noqa: F401

__version__ = "0.1.0"


def register_auto_classes(exist_ok: bool = True):
    """HF Auto-class registration of the MI355X classes (mirrors experiments/robot/openvla_utils.py:38-41); see hf_auto.py."""
    from .hf_auto import register_auto_classes as _r

    return _r(exist_ok)
