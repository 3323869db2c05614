"""emmax -- MI355X-native Emma-X VLA hot path (host-side mirror of prismatic/extern/hf + prismatic/models/vlms)."""

import os as _os

# hipGraph replay of the decode step: ROCm 7.2's default replay path ("graph packet capture") adds ~0.65 us per kernel node on the device;
# with it off the replay runs at the rate of eager launches (profiles/r05_graph_switches.txt).  The runtime reads the variable once, at its
# first HIP call -- so it is set here, at import, UNCONDITIONALLY since round 6 (never overriding the host's own value): replay is then fast
# however graph mode is switched on later (EMMAX_GRAPH=1 or emmax_tuning_set("graph", 1)).  libemmax_hip.so exports the same variable from a
# load-time constructor for hosts that bind the C ABI directly.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

from .config import EmmaXConfig, LlmConfig, TowerConfig  # noqa: F401,E402

__version__ = "0.1.0"


def register_auto_classes(exist_ok: bool = True):
    """HF Auto-class registration of the MI355X classes (mirrors experiments/robot/openvla_utils.py:38-41); see hf_auto.py."""
    from .hf_auto import register_auto_classes as _r

    return _r(exist_ok)
