"""
modeling.py -- `AutoModelForVision2Seq`-compatible model object over the MI355X engine.

Keeps the surface of the reference's HF classes and of the native Emma-X entry point:
  * PrismaticForConditionalGeneration.forward / OpenVLAForActionPrediction.predict_action
        prismatic/extern/hf/modeling_prismatic.py:291-447, 492-566
  * PrismaticVLM.generate_actions(image, prompt_text, type, **kw)        prismatic/models/vlms/prismatic.py:627-696
  * README form  generate_actions(inputs, tokenizer, do_sample=False, max_new_tokens=512) -> (action, reasoning)
        README.md:27-50 (hub-only in the reference; rebuilt from the two sources above)
Same argument meaning and error behaviour; all device math runs in libemmax_hip.so.  The reference is batch-size-1 for
generation (modeling_prismatic.py:460-463); here rows of a batch (<= 64 per GPU) are independent and each equals the
reference's bs=1 result for that row.
"""

from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .actions import ActionTokenizer, token_ids_to_actions, unnormalize
from .config import EmmaXConfig, default_norm_stats
from .engine import EmmaxEngine
from .policy_parser import Solver
from .processing import BatchFeature, EmmaXImageProcessor
from .prompting import PurePromptBuilder
from .tokenizer_stub import StubTokenizer
from .weights import load_hf_state_dict, remap_native_state_dict, synthetic_state_dict, validate_state_dict

PREFIX_TOKEN_ID = 29871   # '▁' appended before action tokens (modeling_prismatic.py:513-516)


def load_tokenizer(path: Optional[str], cfg: EmmaXConfig):
    """The LLaMA-2 tokenizer of a checkpoint directory (tokenizer.json / tokenizer.model), right-padded, capped at
    llm_max_length (prismatic/models/backbones/llm/base_llm.py:146-171); None when the directory holds no tokenizer files."""
    if path and any(os.path.isfile(os.path.join(path, f)) for f in ("tokenizer.json", "tokenizer.model")):
        from transformers import AutoTokenizer   # host side only

        return AutoTokenizer.from_pretrained(path, model_max_length=cfg.llm.max_position, padding_side="right")
    return None


@dataclass
class EmmaXCausalLMOutputWithPast:
    """Field-compatible with PrismaticCausalLMOutputWithPast (modeling_prismatic.py:162-173)."""

    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Optional[Any] = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    attentions: Optional[Tuple[torch.Tensor]] = None
    projector_features: Optional[torch.Tensor] = None


class KVHandle:
    """Opaque stand-in for HF `past_key_values`: the cache lives in the engine's paged KV memory."""

    def __init__(self, engine: EmmaxEngine, lengths: List[int]):
        self.engine, self.lengths = engine, list(lengths)


class EmmaXForActionPrediction:
    config_class = EmmaXConfig

    def __init__(self, config: EmmaXConfig, state_dict: Optional[Dict[str, torch.Tensor]] = None) -> None:
        self.config = config
        self.norm_stats = config.norm_stats
        self.bins = np.linspace(-1, 1, config.n_action_bins)
        self.bin_centers = (self.bins[:-1] + self.bins[1:]) / 2.0
        self.vocab_size = config.llm.vocab_size - config.pad_to_multiple_of   # modeling_prismatic.py:504
        self._state_dict = state_dict
        self.engine: Optional[EmmaxEngine] = None
        self.device = torch.device("cpu")
        self.dtype = torch.bfloat16
        self.training = False
        self.image_transform = EmmaXImageProcessor(config)
        self.tokenizer = None          # set by from_pretrained / from_synthetic, or assign one (`model.tokenizer = ...`)
        self.cache_reserve = 512       # KV room reserved behind a `forward(use_cache=True)` prefill for cached steps

    # ------------------------------------------------------------------------------------------------------------------
    # construction
    # ------------------------------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path: str, torch_dtype: torch.dtype = torch.bfloat16, attn_implementation: Optional[str] = None,
                        low_cpu_mem_usage: bool = True, trust_remote_code: bool = True, load_in_8bit: bool = False,
                        load_in_4bit: bool = False, dtype: Optional[torch.dtype] = None, **_) -> "EmmaXForActionPrediction":
        """HF directory (config.json + *.safetensors [+ dataset_statistics.json]) or a native Prismatic `.pt` file.
        (`dtype=` is the transformers >= 5 spelling of `torch_dtype=`; `config=` passed by the Auto classes is ignored: the
        directory is re-read through EmmaXConfig.from_pretrained.)"""
        if load_in_8bit or load_in_4bit:
            raise NotImplementedError("bitsandbytes quantised loading is outside the MI355X hot path")
        torch_dtype = dtype if dtype is not None else torch_dtype
        if torch_dtype not in (torch.bfloat16, None, "bfloat16", "auto"):
            raise ValueError("the MI355X path computes in bf16 (fp32 accumulate); pass torch_dtype=torch.bfloat16")
        path = str(path)
        if os.path.isfile(path) and path.endswith(".pt"):
            # native Prismatic checkpoint `<run_dir>/checkpoints/<step>.pt` (prismatic/models/load.py:133-144): the run directory
            # must hold `config.json` and `dataset_statistics.json` -- the reference asserts both, and so do we (no made-up
            # un-normalisation statistics)
            ckpt_dir = os.path.dirname(os.path.abspath(path))
            if os.path.basename(ckpt_dir) != "checkpoints":
                raise ValueError(f"Invalid checkpoint! expected `<run_dir>/checkpoints/<name>.pt`, got {path}")
            run_dir = os.path.dirname(ckpt_dir)
            cfg_json, stats_json = os.path.join(run_dir, "config.json"), os.path.join(run_dir, "dataset_statistics.json")
            if not os.path.isfile(cfg_json):
                raise FileNotFoundError(f"Missing `config.json` for run_dir = {run_dir}")
            if not os.path.isfile(stats_json):
                raise FileNotFoundError(f"Missing `dataset_statistics.json` for run_dir = {run_dir}")
            with open(cfg_json) as f:
                raw = json.load(f)
            cfg = EmmaXConfig.from_native_dict(raw)
            with open(stats_json) as f:
                cfg.norm_stats = json.load(f)
            sd = remap_native_state_dict(torch.load(path, map_location="cpu")["model"])
            tok_dir = run_dir
        else:
            cfg = EmmaXConfig.from_pretrained(path)
            sd = load_hf_state_dict(path)
            tok_dir = path
        validate_state_dict(sd, cfg)
        model = cls(cfg, sd)
        model.tokenizer = load_tokenizer(tok_dir, cfg)   # the native class owns its tokenizer (prismatic.py:630); None if no files
        return model

    @classmethod
    def from_synthetic(cls, config: Optional[EmmaXConfig] = None, seed: int = 0, device: str = "cuda:0", planted: bool = False,
                       **engine_kw) -> "EmmaXForActionPrediction":
        """Random-init weights with the real names/shapes, generated directly on `device` (no checkpoint offline)."""
        config = config or EmmaXConfig.emma_x_7b()
        if not config.norm_stats:   # synthetic weights come with synthetic (made-up) statistics; a checkpoint never does
            config.norm_stats = default_norm_stats()
        sd = synthetic_state_dict(config, seed=seed, device=device, dtype=torch.bfloat16, planted=planted)
        m = cls(config, sd)
        m.tokenizer = StubTokenizer()
        return m.to(device, **engine_kw)

    def to(self, device: Union[str, torch.device], max_batch: int = 1, max_prompt: int = 512,
           max_ctx: Optional[int] = None, exact: Optional[bool] = None) -> "EmmaXForActionPrediction":
        """`exact=True`: exact numerics -- the reference's fp32 CPU arithmetic (prismatic.py:659-663 on CPU) instead of bf16 operands
        (include/emmax.h, tuning switch `exact`; 8 rows per launch, larger batches in chunks; None = the library's switch, EMMAX_EXACT)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("EmmaXForActionPrediction runs on MI355X (cuda:N) only; there is no CPU execution path")
        if self.engine is None:
            if self._state_dict is None:
                raise RuntimeError("no weights to load")
            self.engine = EmmaxEngine(self.config, self._state_dict, device=str(device), max_batch=max_batch,
                                      max_prompt=max_prompt, max_ctx=max_ctx, free_state_dict=True, exact=exact)
            self._state_dict = None
        elif device != self.device:
            raise RuntimeError("moving an already-materialised engine between devices is not supported")
        self.device = device
        return self

    def eval(self):
        return self

    def _need_engine(self) -> EmmaxEngine:
        if self.engine is None:
            raise RuntimeError("model is not on a HIP device yet: call .to('cuda:0') (no CPU fallback exists)")
        return self.engine

    # ------------------------------------------------------------------------------------------------------------------
    # stats helpers (modeling_prismatic.py:539-566, prismatic/models/vlas/openvla.py:128-137)
    # ------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _check_unnorm_key(norm_stats: Dict[str, Dict[str, Any]], unnorm_key: Optional[str]) -> str:
        if unnorm_key is None and len(norm_stats) != 1:
            raise ValueError(
                "Your model was trained on more than one dataset. Please pass a `unnorm_key` from the following options to "
                f"choose the statistics used for de-normalizing actions: {norm_stats.keys()}")
        unnorm_key = unnorm_key if unnorm_key is not None else next(iter(norm_stats.keys()))
        if unnorm_key not in norm_stats:
            raise ValueError(f"The `unnorm_key` you chose ({unnorm_key = }) is not in the available statistics. "
                             f"Please choose from: {norm_stats.keys()}")
        return unnorm_key

    def get_action_dim(self, unnorm_key: Optional[str] = None) -> int:
        return len(self.norm_stats[self._check_unnorm_key(self.norm_stats, unnorm_key)]["action"]["q01"])

    def get_action_stats(self, unnorm_key: Optional[str] = None) -> Dict[str, Any]:
        return self.norm_stats[self._check_unnorm_key(self.norm_stats, unnorm_key)]["action"]

    def get_proprio_stats(self, unnorm_key: Optional[str] = None) -> Dict[str, Any]:
        return self.norm_stats[self._check_unnorm_key(self.norm_stats, unnorm_key)]["proprio"]

    def get_prompt_builder(self, system_prompt: Optional[str] = None) -> PurePromptBuilder:
        return PurePromptBuilder("prismatic", system_prompt)

    # ------------------------------------------------------------------------------------------------------------------
    # core: vision + prefill (+ greedy loop)
    # ------------------------------------------------------------------------------------------------------------------
    def _rows(self, input_ids, attention_mask=None) -> List[List[int]]:
        """[B,P] tensor (+ optional right-padding mask) or ragged lists -> list of per-row id lists."""
        if isinstance(input_ids, torch.Tensor):
            ids = input_ids.detach().to("cpu", torch.long)
            if ids.ndim == 1:
                ids = ids[None]
            if attention_mask is not None:
                lens = attention_mask.detach().to("cpu").long().sum(-1).tolist()
            else:
                lens = [ids.shape[1]] * ids.shape[0]
            return [ids[b, : lens[b]].tolist() for b in range(ids.shape[0])]
        return [list(map(int, r)) for r in input_ids]

    def _encode_images(self, pixel_values=None, frames_u8=None) -> torch.Tensor:
        """`pixel_values` is the model input (reference contract).  `frames_u8` (the uint8 frame the processor ALSO emits) is
        only a faster route to the same numbers -- normalisation fused into the patch gather -- and is used when it is given
        alone, or when the caller's `pixel_values` still carries the processor's tag for that very frame (an augmented /
        re-cropped `pixel_values` tensor is a new object without the tag and is what gets encoded)."""
        eng = self._need_engine()
        if frames_u8 is not None and (pixel_values is None or getattr(pixel_values, "_emmax_frames_tag", None) is getattr(frames_u8, "_emmax_frames_tag", 0)):
            return eng.vision_encode(frames_u8.to(self.device).contiguous())
        if isinstance(pixel_values, dict):   # native layout {"dino": [B,3,H,W], "siglip": [B,3,H,W]}
            pixel_values = torch.cat([pixel_values["dino"], pixel_values["siglip"]], dim=1)
        return eng.vision_encode_pixels(pixel_values)

    def _prefill(self, rows: List[List[int]], pixel_values=None, frames_u8=None, max_new: int = 0) -> torch.Tensor:
        eng = self._need_engine()
        B = len(rows)
        src = pixel_values if pixel_values is not None else frames_u8
        nimg = src["dino"].shape[0] if isinstance(src, dict) else src.shape[0]
        if nimg != B:
            raise ValueError("Non-homogenous batch of (text, image) input -- forward() does not support mixed batches!")
        P = max(len(r) for r in rows)
        if P > self.config.llm.max_position:
            rows = [r[: self.config.llm.max_position] for r in rows]   # tokenizer truncation at llm_max_length
            P = self.config.llm.max_position
        eng.ensure_capacity(B, P, max(max_new, 1))
        patches = self._encode_images(pixel_values, frames_u8)
        eng.prefill(rows, patches)
        return patches

    def forward(self, input_ids=None, attention_mask=None, pixel_values=None, labels=None, inputs_embeds=None,
                past_key_values=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                output_projector_features=None, return_dict=None, frames_u8=None):
        """Multimodal forward -> logits for every position ([B,S,V] when rows have equal length, else a list), or one
        cached decode step when `input_ids.shape[1] == 1` and `past_key_values` is the handle of a previous call."""
        eng = self._need_engine()
        if labels is not None:
            raise NotImplementedError("loss / labels belong to training, outside the inference hot path")
        if output_attentions or output_hidden_states:
            raise NotImplementedError("attention maps / hidden states are not materialised by the fused kernels")
        # `inputs_embeds` is never consumed as data by the reference either: the cached branch hands the language model
        # inputs_embeds=None (modeling_prismatic.py:330-341), the multimodal branch embeds `input_ids` whatever else was passed
        # (:381), and the language-only branch asserts it away (:345).  Same behaviour here.
        if input_ids is None:
            raise ValueError("forward() needs `input_ids`: every branch of the reference embeds them (`inputs_embeds` is not read)")
        rows = self._rows(input_ids, attention_mask)
        if isinstance(input_ids, torch.Tensor) and input_ids.shape[-1] == 1 and past_key_values is not None:
            assert len(rows) == len(past_key_values.lengths), "cached step must keep the batch of the prefill"
            eng.set_current_tokens([r[0] for r in rows])
            eng.decode_step()
            logits = eng.last_logits()[:, None, :]
            past_key_values.lengths = [n + 1 for n in past_key_values.lengths]
            return EmmaXCausalLMOutputWithPast(logits=logits, past_key_values=past_key_values)
        if input_ids is not None and isinstance(input_ids, torch.Tensor) and input_ids.shape[-1] == 1:
            assert past_key_values is not None, "You must provide `past_key_values` during cached generation!"
        if pixel_values is None and frames_u8 is None:
            # unimodal forward (modeling_prismatic.py:343-359): text only, no patch rows
            assert inputs_embeds is None, "Missing `input_ids` in language-only forward!"
            assert past_key_values is None, "Unexpected key `past_key_values` provided during language-only forward!"
            eng.ensure_capacity(len(rows), max(len(r) for r in rows), self.cache_reserve if use_cache else 1)
            eng.prefill(rows, None)
            per_row = eng.prefill_logits()
            same = len({t.shape[0] for t in per_row}) == 1
            return EmmaXCausalLMOutputWithPast(logits=torch.stack(per_row) if same else per_row,
                                               past_key_values=KVHandle(eng, eng._last_S) if use_cache else None)
        patches = self._prefill(rows, pixel_values, frames_u8, max_new=self.cache_reserve if use_cache else 0)
        per_row = eng.prefill_logits()
        same = len({t.shape[0] for t in per_row}) == 1
        logits = torch.stack(per_row) if same else per_row
        return EmmaXCausalLMOutputWithPast(
            logits=logits, past_key_values=KVHandle(eng, eng._last_S) if use_cache else None,
            projector_features=patches if output_projector_features else None)

    __call__ = forward

    @torch.inference_mode()
    def generate_ids(self, rows: List[List[int]], pixel_values=None, frames_u8=None, max_new_tokens: int = 512,
                     stop_on_eos: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        """Greedy decode for every row. Returns device tensors (new_ids int32 [B,max_new] pad-filled, lens int32 [B])."""
        eng = self._need_engine()
        self._prefill(rows, pixel_values, frames_u8, max_new=max_new_tokens)
        return eng.generate(max_new_tokens, stop_on_eos)

    def _max_new(self, rows, max_new_tokens, max_length, min_length, had_max_new: bool = True) -> int:
        """HF length semantics: `max_length` counts the PROMPT too (as HF counts it: the text ids, not the patch rows) and only
        applies when `max_new_tokens` is not given; `min_length` (total length, HF default 0) above the prompt length would
        suppress EOS -- outside the greedy hot path, so it raises instead of being ignored."""
        plen = max(len(r) for r in rows)
        if min_length is not None and int(min_length) > plen + 1:
            raise NotImplementedError(f"min_length={min_length} beyond the prompt ({plen} ids) needs EOS suppression, outside the hot path")
        if max_new_tokens is None:
            if max_length is None:
                max_new_tokens = 20       # HF GenerationConfig default max_length=20 is a total; keep the historical default here
            else:
                max_new_tokens = max(int(max_length) - plen, 1)
        return int(max_new_tokens)

    @torch.inference_mode()
    def generate(self, input_ids=None, pixel_values=None, attention_mask=None, max_new_tokens: Optional[int] = None, do_sample: bool = False,
                 min_length: int = 1, temperature: float = 0.0, frames_u8=None, **kwargs) -> torch.Tensor:
        """HF-style: returns LongTensor [B, P + T] = prompt ++ generated (right-padded with pad_token_id)."""
        if do_sample:
            raise NotImplementedError("only greedy decoding (do_sample=False) is on the hot path")
        if kwargs.get("num_beams", 1) != 1:
            raise NotImplementedError("beam search is outside the hot path")
        rows = self._rows(input_ids, attention_mask)
        max_new_tokens = self._max_new(rows, max_new_tokens, kwargs.get("max_length"), min_length)
        new_ids, lens = self.generate_ids(rows, pixel_values, frames_u8, max_new_tokens)
        new_ids, lens = new_ids.cpu(), lens.cpu().tolist()
        T = max(lens)
        P = max(len(r) for r in rows)
        out = torch.full((len(rows), P + T), self.config.pad_token_id, dtype=torch.long)
        for b, r in enumerate(rows):
            seq = r + new_ids[b, : lens[b]].tolist()
            out[b, : len(seq)] = torch.tensor(seq, dtype=torch.long)
        return out

    # ------------------------------------------------------------------------------------------------------------------
    # action APIs
    # ------------------------------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def predict_action(self, input_ids=None, unnorm_key: Optional[str] = None, **kwargs) -> np.ndarray:
        """modeling_prismatic.py:506-537: append 29871 if missing, generate `action_dim` tokens, de-tokenise, un-normalise."""
        rows = self._rows(input_ids, kwargs.pop("attention_mask", None))
        if len(rows) != 1:
            raise ValueError("Generation with batch size > 1 is not currently supported!")   # reference contract
        if rows[0][-1] != PREFIX_TOKEN_ID:
            rows[0] = rows[0] + [PREFIX_TOKEN_ID]
        dim = self.get_action_dim(unnorm_key)
        new_ids, lens = self.generate_ids(rows, kwargs.get("pixel_values"), kwargs.get("frames_u8"), max_new_tokens=dim)
        full = rows[0] + new_ids[0, : int(lens[0])].cpu().tolist()
        predicted = np.array(full[-dim:])
        normalized = token_ids_to_actions(predicted, self.vocab_size, self.bin_centers)
        return unnormalize(normalized, self.get_action_stats(unnorm_key))

    def _solver(self, tokenizer) -> Solver:
        return Solver(ActionTokenizer(tokenizer, bins=self.config.n_action_bins), verbose=False)

    def _postprocess(self, generated_ids: List[int], tokenizer, type: str = "act"):
        """prismatic.py:666-696: decode -> Solver -> un-normalise."""
        text = tokenizer.decode(generated_ids, skip_special_tokens=True).strip()
        s = self._solver(tokenizer)
        if type == "act":
            actions, _ = s.extract_action_policies(text)
            stats = self.get_action_stats(None)
            return [unnormalize(np.array(a), stats) for a in actions], text
        if type == "pos":
            require_unorm, delta = s.extract_movement_plan(text)
            proprio = delta
            if require_unorm:
                proprio = unnormalize(np.array(delta), self.get_proprio_stats(), "Q1", "Q99")
            return proprio, text
        raise ValueError(f"Unsupported generate_actions type `{type}` (expected 'act' or 'pos')")

    @torch.inference_mode()
    def generate_actions(self, *args, **kwargs):
        """Two call forms, positional or by keyword:
          README   generate_actions(inputs, tokenizer, do_sample=False, max_new_tokens=512) -> (action[7], reasoning)
          native   generate_actions(image, prompt_text, type, **generate_kwargs) -> (list of action[7] | proprio, generated_text)
                   (prismatic/models/vlms/prismatic.py:628; called as `vla.generate_actions(image=image, prompt_text=prompt,
                   type=type, temperature=0.0, max_new_tokens=512, min_length=1, do_sample=False)` by
                   experiments/robot/openvla_utils.py:215-217).  The tokenizer is the model's own (`self.tokenizer`, attached by
                   from_pretrained like the reference's `llm_backbone.tokenizer`); `tokenizer=` overrides it.
        """
        args = list(args)
        inputs = kwargs.pop("inputs", None)
        if inputs is None and args and hasattr(args[0], "keys") and "input_ids" in args[0]:
            inputs = args.pop(0)   # dict, our BatchFeature, or transformers.BatchFeature (a UserDict: processing_prismatic.py:216)
        if kwargs.get("do_sample", False):
            raise NotImplementedError("only greedy decoding (do_sample=False) is on the hot path")
        if kwargs.get("num_beams", 1) != 1:
            raise NotImplementedError("beam search is outside the hot path")
        if inputs is not None:   # README form
            tokenizer = args.pop(0) if args else kwargs.pop("tokenizer", None)
            tokenizer = tokenizer if tokenizer is not None else self.tokenizer
            if tokenizer is None:
                raise ValueError("generate_actions(inputs, tokenizer): no tokenizer given and none attached to the model")
            rows = self._rows(inputs["input_ids"], inputs.get("attention_mask"))
            if len(rows) != 1:
                raise ValueError("Generation with batch size > 1 is not currently supported!")
            max_new = self._max_new(rows, kwargs.get("max_new_tokens", None if "max_length" in kwargs else 512), kwargs.get("max_length"),
                                    kwargs.get("min_length"))
            new_ids, lens = self.generate_ids(rows, inputs.get("pixel_values"), inputs.get("frames_u8"), max_new_tokens=max_new)
            actions, text = self._postprocess(new_ids[0, : int(lens[0])].cpu().tolist(), tokenizer, "act")
            return actions[0], text
        # native form
        names = ("image", "prompt_text", "type")
        vals = {}
        for n in names:
            if args:
                if n in kwargs:
                    raise TypeError(f"generate_actions() got multiple values for argument '{n}'")
                vals[n] = args.pop(0)
            elif n in kwargs:
                vals[n] = kwargs.pop(n)
            else:
                raise TypeError(f"generate_actions() missing required argument: '{n}'")
        if args:
            raise TypeError(f"generate_actions() takes 3 positional arguments but {3 + len(args)} were given")
        tokenizer = kwargs.pop("tokenizer", None)
        tokenizer = tokenizer if tokenizer is not None else self.tokenizer
        if tokenizer is None:
            raise ValueError("no tokenizer attached: the checkpoint directory holds no tokenizer files; set model.tokenizer")
        if vals["type"] not in ("act", "pos"):
            raise ValueError(f"Unsupported generate_actions type `{vals['type']}` (expected 'act' or 'pos')")
        enc = tokenizer(vals["prompt_text"], truncation=True, return_tensors="pt")
        rows = self._rows(enc.input_ids)
        max_new = self._max_new(rows, kwargs.get("max_new_tokens"), kwargs.get("max_length"), kwargs.get("min_length"))
        feat = self.image_transform(vals["image"])
        new_ids, lens = self.generate_ids(rows, feat["pixel_values"], feat.get("frames_u8"), max_new_tokens=max_new)
        return self._postprocess(new_ids[0, : int(lens[0])].cpu().tolist(), tokenizer, vals["type"])

    @torch.inference_mode()
    def generate_actions_batch(self, frames_u8: torch.Tensor, prompt_rows: Sequence[Sequence[int]], max_new_tokens: int = 512,
                               stop_on_eos: bool = True, tokenizer=None):
        """Batched extension: returns (actions f32 [B,7], new_ids int32 [B,T], lens int32 [B]).

        With `tokenizer` each row goes ids -> text -> Solver exactly like the bs=1 path; without it the ids-level
        stand-in `actions_from_ids` is used (synthetic weights / throughput runs)."""
        new_ids, lens = self.generate_ids([list(r) for r in prompt_rows], None, frames_u8, max_new_tokens, stop_on_eos)
        ids_h, lens_h = new_ids.cpu(), lens.cpu().tolist()
        acts = np.zeros((len(lens_h), 7), dtype=np.float32)
        stats = self.get_action_stats(None)
        for b, n in enumerate(lens_h):
            row = ids_h[b, :n].tolist()
            if tokenizer is not None:
                a, _ = self._postprocess(row, tokenizer, "act")
                acts[b] = a[0]
            else:
                acts[b] = self.actions_from_ids(row, stats)
        return acts, new_ids, lens

    def actions_from_ids(self, row: List[int], stats: Dict[str, Any]) -> np.ndarray:
        """ids-level stand-in for Solver.extract_action_policies when no tokenizer round trip is wanted (synthetic
        runs / throughput benches): the first 7 ids of the LAST run of >= 7 action-range ids (the policy group is the
        last thing the grammar emits: "...MOVEMENT:\n..\nPOLICIES:\n{tokens}\n"); zeros when there is none, like the
        reference's parse-failure contract.  With a real checkpoint use the tokenizer path (default)."""
        lo = self.vocab_size - self.config.n_action_bins
        runs: List[List[int]] = []
        cur: List[int] = []
        for t in row:
            if lo <= t < self.vocab_size:
                cur.append(t)
            else:
                if cur:
                    runs.append(cur)
                cur = []
        if cur:
            runs.append(cur)
        runs = [r for r in runs if len(r) >= 7]
        if not runs:
            return np.zeros(7, dtype=np.float32)
        seven = runs[-1][:7]
        return unnormalize(token_ids_to_actions(np.array(seven), self.vocab_size, self.bin_centers), stats).astype(np.float32)
