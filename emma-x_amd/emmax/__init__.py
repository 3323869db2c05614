"""emmax -- MI355X-native Emma-X VLA hot path (host-side mirror of prismatic/extern/hf + prismatic/models/vlms)."""

import os as _os

# hipGraph replay of the decode step (EMMAX_GRAPH=1): ROCm 7.2's default replay path ("graph packet capture") adds ~0.65 us per kernel node on
# the device; with it off the replay runs at the rate of eager launches (profiles/r05_graph_switches.txt).  The runtime reads the variable once,
# at its first HIP call -- so it is set here, at import, and only when the host asked for graph replay through the environment; a host that
# switches replay on later (emmax_tuning_set("graph", 1)) exports it itself before its first HIP call.
if _os.environ.get("EMMAX_GRAPH") == "1":
    _os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

from .config import EmmaXConfig, LlmConfig, TowerConfig  # noqa: F401,E402

__version__ = "0.1.0"


def register_auto_classes(exist_ok: bool = True):
    """HF Auto-class registration of the MI355X classes (mirrors experiments/robot/openvla_utils.py:38-41); see hf_auto.py."""
    from .hf_auto import register_auto_classes as _r

    return _r(exist_ok)
