"""
prompting.py -- prompt construction for the `llama2-7b-pure` backbone.

Behavioural mirror of `PurePromptBuilder` (prismatic/models/backbones/llm/prompting/base_prompter.py:28-73) and of the
Bridge task template used by the callers (experiments/robot/bridge/run_bridgev2_eval.py:167-168).  Strings are pinned
against the reference in tests/golden/prompts.json.
"""

from __future__ import annotations

from typing import Optional, Sequence

BOS, EOS = "<s>", "</s>"


class PurePromptBuilder:
    """Alternating human/gpt turns rendered as `In: ...\\nOut: ...</s>`."""

    def __init__(self, model_family: str = "prismatic", system_prompt: Optional[str] = None) -> None:
        self.model_family, self.system_prompt = model_family, system_prompt
        self.bos, self.eos = BOS, EOS
        self.prompt, self.turn_count = "", 0

    @staticmethod
    def _human(msg: str) -> str:
        return f"In: {msg}\nOut: "

    def _gpt(self, msg: str) -> str:
        return (msg if msg != "" else " ") + self.eos

    def add_turn(self, role: str, message: str) -> str:
        expected = "human" if self.turn_count % 2 == 0 else "gpt"
        assert role == expected, f"turn {self.turn_count} must come from `{expected}`"
        text = message.replace("<image>", "").strip()
        piece = self._human(text) if role == "human" else self._gpt(text)
        self.prompt += piece
        self.turn_count += 1
        return piece

    def get_potential_prompt(self, message: str) -> str:
        return (self.prompt + self._human(message)).removeprefix(self.bos).rstrip()

    def get_prompt(self) -> str:
        # the tokenizer inserts <s> itself
        return self.prompt.removeprefix(self.bos).rstrip()


def bridge_task_label(instruction: str, gripper_xy: Optional[Sequence[int]] = None) -> str:
    label = f"What action should the robot take to achieve the instruction\nINSTRUCTION: \n{instruction}\n"
    if gripper_xy is not None:
        label += f"CURRENT GRIPPER: [{gripper_xy[0]}, {gripper_xy[1]}]\n"
    return label


def build_prompt(task_label: str) -> str:
    b = PurePromptBuilder("prismatic")
    b.add_turn("human", task_label)
    return b.get_prompt()
