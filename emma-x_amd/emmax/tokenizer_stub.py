"""
tokenizer_stub.py -- deterministic character-level stand-in for the LLaMA-2 tokenizer.

The reference loads `meta-llama/Llama-2-7b-hf`'s tokenizer from the hub (prismatic/vla/solver.py:188-190,
prismatic/models/backbones/llm/llama2.py:55-76); those files do not exist offline, so synthetic runs and the tests use
this stub.  It keeps the properties the hot path relies on:
  * vocab_size 32000, <unk>/<s>/</s> = 0/1/2, BOS auto-prepended when add_special_tokens=True,
  * a dummy-prefix token 29871 (`▁`) in front of every encoded piece -- the "first token is meaningless" that
    Solver drops (solver.py:125-126) and that predict_action appends (modeling_prismatic.py:513-516),
  * the last 256 ids (31744..31999) are single printable characters, so action tokens survive ids -> text -> ids.
When a real checkpoint directory holds tokenizer files, `EmmaXProcessor.from_pretrained` loads those instead.
"""

from __future__ import annotations

from typing import List, Sequence, Union

import torch

PREFIX_ID = 29871


class _Encoding(dict):
    """Tiny BatchEncoding look-alike: attribute + key access."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class StubTokenizer:
    vocab_size = 32000
    unk_token_id, bos_token_id, eos_token_id = 0, 1, 2
    pad_token_id = 32000
    padding_side = "right"
    model_max_length = 2048

    # id <-> char:  3..258 = code points 0..255;  259..31999 (minus 29871) = U+4E00 + (id - 259)
    @staticmethod
    def _char_to_id(c: str) -> int:
        o = ord(c)
        if o < 256:
            return 3 + o
        i = o - 0x4E00 + 259
        if 259 <= i < 32000 and i != PREFIX_ID:
            return i
        return 0

    @staticmethod
    def _id_to_char(i: int) -> str:
        if 3 <= i < 259:
            return chr(i - 3)
        if 259 <= i < 32000 and i != PREFIX_ID:
            return chr(0x4E00 + i - 259)
        return ""

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        ids = [self.bos_token_id] if add_special_tokens else []
        ids.append(PREFIX_ID)
        ids.extend(self._char_to_id(c) for c in text)
        return ids

    def __call__(self, text: Union[str, Sequence[str]], add_special_tokens: bool = True, truncation: bool = False,
                 max_length: int = None, padding: bool = False, return_tensors: str = None, **_):
        single = isinstance(text, str)
        rows = [self.encode(t, add_special_tokens) for t in ([text] if single else text)]
        if truncation:
            lim = max_length or self.model_max_length
            rows = [r[:lim] for r in rows]
        if return_tensors == "pt":
            n = max(len(r) for r in rows)
            if any(len(r) != n for r in rows) and not padding:
                raise ValueError("Unable to create tensor: rows have different lengths and padding=False")
            ids = torch.full((len(rows), n), self.pad_token_id, dtype=torch.long)
            mask = torch.zeros((len(rows), n), dtype=torch.long)
            for i, r in enumerate(rows):
                ids[i, : len(r)] = torch.tensor(r)
                mask[i, : len(r)] = 1
            return _Encoding(input_ids=ids, attention_mask=mask)
        if single:
            return _Encoding(input_ids=rows[0], attention_mask=[1] * len(rows[0]))
        return _Encoding(input_ids=rows, attention_mask=[[1] * len(r) for r in rows])

    def decode(self, ids, skip_special_tokens: bool = False, **_) -> str:
        if isinstance(ids, torch.Tensor):
            ids = ids.tolist()
        out = []
        for i in ids:
            i = int(i)
            if i in (0, 1, 2) or i >= 32000:
                if not skip_special_tokens:
                    out.append({0: "<unk>", 1: "<s>", 2: "</s>"}.get(i, "<PAD>"))
                continue
            out.append(self._id_to_char(i))
        return "".join(out)

    def batch_decode(self, seqs, skip_special_tokens: bool = False, **kw) -> List[str]:
        return [self.decode(s, skip_special_tokens=skip_special_tokens, **kw) for s in seqs]
