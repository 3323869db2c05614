"""
actions.py -- 7-DoF action <-> token-id mapping on the last 256 vocabulary ids, and un-normalisation.

Behavioural mirror of `ActionTokenizer` (prismatic/vla/action_tokenizer.py:13-72) and of the tail of
`OpenVLAForActionPrediction.predict_action` (prismatic/extern/hf/modeling_prismatic.py:522-535).  Host-side integer and
fp64 numpy math; bit-exact against the reference (tests/golden/action_decode.npz).
"""

from __future__ import annotations

from typing import Any, Dict, List, Sequence, Union

import numpy as np


class ActionTokenizer:
    """Uniform `bins`-level discretiser that re-uses the `bins` least-used (= last) token ids of the base tokenizer."""

    def __init__(self, tokenizer, bins: int = 256, min_action: int = -1, max_action: int = 1) -> None:
        self.tokenizer, self.n_bins = tokenizer, bins
        self.min_action, self.max_action = min_action, max_action
        edges = np.linspace(min_action, max_action, bins)
        self.bins = edges
        self.bin_centers = (edges[:-1] + edges[1:]) / 2.0
        self.action_token_begin_idx: int = int(tokenizer.vocab_size - (bins + 1))

    # continuous -> text (what the training data contains)
    def __call__(self, action: np.ndarray) -> Union[str, List[str]]:
        ids = self.encode_ids(action)
        if ids.ndim == 1:
            return self.tokenizer.decode(list(ids))
        return self.tokenizer.batch_decode(ids.tolist())

    def encode_ids(self, action: np.ndarray) -> np.ndarray:
        """continuous -> token ids (no text round trip)."""
        clipped = np.clip(action, a_min=float(self.min_action), a_max=float(self.max_action))
        return self.tokenizer.vocab_size - np.digitize(clipped, self.bins)

    # token ids -> bin centres
    def decode_token_ids_to_actions(self, action_token_ids: np.ndarray) -> np.ndarray:
        return token_ids_to_actions(action_token_ids, self.tokenizer.vocab_size, self.bin_centers)

    @property
    def vocab_size(self) -> int:
        return self.n_bins


def token_ids_to_actions(ids: np.ndarray, vocab_size: int, centers: np.ndarray) -> np.ndarray:
    """id -> bin index (vocab_size - id - 1, clamped to the valid centre range) -> centre value."""
    idx = np.clip(vocab_size - np.asarray(ids) - 1, a_min=0, a_max=centers.shape[0] - 1)
    return centers[idx]


def unnormalize(normalized: np.ndarray, stats: Dict[str, Any], lo_key: str = "q01", hi_key: str = "q99") -> np.ndarray:
    """Map [-1,1] back to the dataset range on masked dimensions (gripper dimension passes through)."""
    lo, hi = np.array(stats[lo_key]), np.array(stats[hi_key])
    mask = stats.get("mask", np.ones_like(stats[lo_key], dtype=bool))
    return np.where(mask, 0.5 * (np.asarray(normalized) + 1) * (hi - lo) + lo, normalized)
