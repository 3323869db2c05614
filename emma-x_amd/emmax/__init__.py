"""emmax -- MI355X-native Emma-X VLA hot path (host-side mirror of prismatic/extern/hf + prismatic/models/vlms)."""

from .config import EmmaXConfig, LlmConfig, TowerConfig  # noqa: F401

__version__ = "0.1.0"
