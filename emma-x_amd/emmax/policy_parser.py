"""
policy_parser.py -- generated grounded-chain-of-thought text -> 7-DoF policies / movement plan.

Behavioural mirror of `Solver` (prismatic/vla/solver.py:8-137): same keys, same fall-backs (a malformed policy line
yields `[[0]*7]` and the untouched text; a malformed movement line yields `[-100]*7`), never raises.  Pinned against the
reference class in tests/golden/solver.json.
"""

from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import ast

import numpy as np

from .actions import ActionTokenizer

POLICY_KEY, MOVEMENT_KEY, GRIPPER_KEY = "POLICIES:", "MOVEMENT:", "NEXT GRIPPER:"

# textual movement vocabulary: phrase -> (sign, axis)
_DIRECTIONS: Dict[str, Tuple[int, str]] = {}
for _verb, _axis, _neg, _pos in (
    ("move", "y", "backward", "forward"), ("move", "x", "right", "left"), ("move", "z", "downward", "upward"),
    ("roll", "ox", "downward", "upward"), ("swing", "ox", "downward", "upward"), ("pitch", "oy", "downward", "upward"),
    ("yaw", "oz", "downward", "upward"), ("rotate", "oz", "clockwise", "counterclockwise"),
):
    _DIRECTIONS[f"{_verb}_{_neg}"] = (-1, _axis)
    _DIRECTIONS[f"{_verb}_{_pos}"] = (1, _axis)
_DIRECTIONS["close_gripper"] = (-1, "grip")
_DIRECTIONS["open_gripper"] = (1, "grip")
_AXES = ("x", "y", "z", "ox", "oy", "oz", "grip")


def _first_content_line(block: str) -> str:
    lines = [ln for ln in block.split("\n") if ln.strip()]
    return lines[0].strip()   # IndexError on an empty block is part of the contract (caught by the callers)


class Solver:
    def __init__(self, action_tokenizer: Optional[ActionTokenizer] = None, verbose: bool = True) -> None:
        self.action_tokenizer, self.verbose = action_tokenizer, verbose
        self.coordinates_key, self.movement_key, self.policy_key = GRIPPER_KEY, MOVEMENT_KEY, POLICY_KEY

    def _ids_to_actions(self, text: str) -> np.ndarray:
        ids = self.action_tokenizer.tokenizer(text, add_special_tokens=False).input_ids
        return self.action_tokenizer.decode_token_ids_to_actions(np.array(ids))

    # ---- POLICIES: <8 tokens>[;<8 tokens>...] ----
    def extract_action_policies(self, text: str):
        try:
            if self.policy_key in text:
                cut = text.index(self.policy_key)
                remain_text = text[:cut]
                line = _first_content_line(text[cut + len(self.policy_key):])
            else:
                remain_text, line = "", text.strip()
            policies = []
            for group in line.split(";"):
                vec = self._ids_to_actions(group)[1:8]      # token 0 is the tokenizer's dummy prefix
                if len(vec) != 7:
                    # the reference falls into its catch-all here (list has no .tolist()): whole result -> zeros
                    raise ValueError("policy group does not decode to 7 dimensions")
                policies.append(vec.tolist())
        except Exception:
            policies, remain_text = [[0] * 7], text
        return policies, remain_text

    # ---- MOVEMENT: tokenised (normalised) or textual ("move forward 3; ...") ----
    def extract_movement_plan(self, text: str):
        require_unorm = None
        try:
            line = _first_content_line(text[text.index(self.movement_key) + len(self.movement_key):])
            if "gripper" not in line:
                require_unorm = True
                movement = self._ids_to_actions(line)[1:8]
                assert len(movement) == 7
            else:
                require_unorm = False
                totals = dict.fromkeys(_AXES, 0)
                for phrase in [s for s in line.split(";") if len(s) > 0][:7]:
                    words = phrase.split()
                    sign, axis = _DIRECTIONS["_".join(words[:2])]
                    if axis == "grip":
                        unit, level = 1, round("open" in phrase)
                    else:
                        unit = 1e-3 if axis.startswith("o") else 1 / 180 * np.pi
                        level = int(words[2])
                    totals[axis] += sign * unit * level
                movement = [totals[a] for a in _AXES]
        except Exception:
            movement = [-100] * 7
        return require_unorm, np.array(movement)

    def extract_2d_coordinates(self, text: str):
        """The first non-empty line after "NEXT GRIPPER:" as a Python value -- the reference `eval`s it (solver.py:33-40), so lists,
        tuples, floats and arithmetic all pass and come back as whatever they evaluate to; anything else (and every failure) is
        [0, 0].  Generated text is untrusted: the line is evaluated over its literal / arithmetic syntax tree only -- names and
        calls, which the reference would execute, give [0, 0] here (an undefined name gives [0, 0] in the reference too)."""
        try:
            block = text[text.index(self.coordinates_key) + len(self.coordinates_key):]
            return _safe_eval(_first_content_line(block))
        except Exception:
            return [0, 0]


_BIN = {ast.Add: lambda a, b: a + b, ast.Sub: lambda a, b: a - b, ast.Mult: lambda a, b: a * b, ast.Div: lambda a, b: a / b,
        ast.FloorDiv: lambda a, b: a // b, ast.Mod: lambda a, b: a % b}


def _safe_eval(src: str):
    """`eval` restricted to literals, tuples / lists and + - * / // % on numbers."""
    def ev(n):
        if isinstance(n, ast.Constant) and (n.value is None or isinstance(n.value, (int, float, complex, str, bool))):
            return n.value
        if isinstance(n, ast.List):
            return [ev(e) for e in n.elts]
        if isinstance(n, ast.Tuple):
            return tuple(ev(e) for e in n.elts)
        if isinstance(n, ast.UnaryOp) and isinstance(n.op, (ast.UAdd, ast.USub)):
            v = ev(n.operand)
            return +v if isinstance(n.op, ast.UAdd) else -v
        if isinstance(n, ast.BinOp) and type(n.op) in _BIN:
            a, b = ev(n.left), ev(n.right)
            if not all(isinstance(v, (int, float, complex)) and not isinstance(v, bool) for v in (a, b)):
                raise ValueError("arithmetic on non-numbers")
            return _BIN[type(n.op)](a, b)
        raise ValueError("outside the literal / arithmetic subset")

    return ev(ast.parse(src.strip(), mode="eval").body)
