"""
engine.py -- thin Python owner of the libemmax_hip handles: device memory (torch tensors), weight binding, sessions.

PyTorch is used for plumbing only (device allocations, the current HIP stream); every FLOP on the hot path happens in
the hand-written gfx950 kernels behind the C ABI (include/emmax.h).
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from .config import EmmaXConfig
from .weights import param_shapes, validate_state_dict


def _config_c(cfg: EmmaXConfig) -> _lib.ConfigC:
    c = _lib.ConfigC()
    for i, tw in enumerate(cfg.towers):
        t = c.tower[i]
        t.embed_dim, t.depth, t.num_heads, t.mlp_hidden = tw.embed_dim, tw.depth, tw.num_heads, tw.mlp_hidden
        t.has_cls, t.n_reg, t.layerscale = int(tw.has_cls), tw.n_reg, int(tw.layerscale)
        t.patch, t.image_size, t.take_index = tw.patch, tw.image_size, tw.take_index
        t.ln_eps = tw.ln_eps
        for j in range(3):
            t.mean[j] = tw.mean[j]
            t.std[j] = tw.std[j]
    L = cfg.llm
    c.hidden, c.inter, c.n_layers, c.n_heads, c.n_kv_heads, c.head_dim, c.vocab = (
        L.hidden_size, L.intermediate_size, L.num_layers, L.num_heads, L.num_kv_heads, L.head_dim, L.vocab_size)
    c.rms_eps, c.rope_theta = L.rms_eps, L.rope_theta
    c.bos_id, c.eos_id, c.pad_id = cfg.bos_token_id, cfg.eos_token_id, cfg.pad_token_id
    c.decode_fp8 = 1 if getattr(cfg, "decode_weight_dtype", "bf16") == "fp8" else 0
    return c


class _Staged:
    """A staged prefill in flight on the admission stream."""

    def __init__(self, n, event, keep):
        self.n, self.event, self.keep, self.committed = n, event, keep, 0

    def ready(self) -> bool:
        return self.event.query()

    def wait(self) -> None:
        self.event.synchronize()


def _device_bytes(nbytes: int, device, what: str):
    """The weight arenas ("ARENA", "AUX") and the paged KV region ("KV") come from torch's allocator (tools/malloc_flags_probe.py replaces
    this function to measure uncached / fine-grained device memory: no difference, profiles/r05_malloc_flags_ab.txt)."""
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


class EmmaxEngine:
    """Model weights (re-laid-out into one device arena) + one session (workspace + paged KV cache)."""

    def __init__(self, cfg: EmmaXConfig, state_dict: Dict[str, torch.Tensor], device: str = "cuda:0",
                 max_batch: int = 1, max_prompt: int = 512, max_ctx: Optional[int] = None, free_state_dict: bool = False,
                 exact: Optional[bool] = None):
        self.lib = _lib.load()
        # EXACT NUMERICS (include/emmax.h, tuning switch `exact`): the fp32 CPU arithmetic of the reference instead of bf16 operands.  Frozen into
        # the model at finalize (ViT LayerNorms unfolded) and into every session of this engine; None = what the library's switch says (EMMAX_EXACT)
        # (1 / True: 24-bit K / V cache, the default exact format; 2: fp32 cache, its A/B partner)
        self.exact = int(_lib.tuning_get("exact")) if exact is None else int(exact)
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.EmmaxError("EmmaxEngine needs a HIP device (cuda:N); the product path has no CPU fallback")
        validate_state_dict(state_dict, cfg)
        torch.cuda.set_device(self.device)
        self._model = C.c_void_p()
        self._session = C.c_void_p()
        cc = _config_c(cfg)
        _lib.check(self.lib.emmax_model_create(C.byref(cc), C.byref(self._model)), "emmax_model_create")
        # ---- bind (bf16, on device) and finalize into the arena ----
        keep: List[torch.Tensor] = []
        for key, shape, _ in param_shapes(cfg):
            t = state_dict[key]
            if t.device != self.device or t.dtype != torch.bfloat16 or not t.is_contiguous():
                t = t.to(device=self.device, dtype=torch.bfloat16).contiguous()
            keep.append(t)
            shp = (C.c_int64 * t.ndim)(*t.shape)
            _lib.check(self.lib.emmax_model_bind_weight(self._model, key.encode(), t.data_ptr(), 0, shp, t.ndim),
                       f"bind {key}")
            if free_state_dict:
                state_dict[key] = None  # drop the caller's copy as soon as ours exists
        nbytes = self.lib.emmax_model_arena_bytes(self._model)
        self.arena = _device_bytes(nbytes, self.device, "ARENA")
        stream = _lib.current_stream()
        import time

        t0 = time.perf_counter()
        with _lib.tuning(exact=int(self.exact)):
            _lib.check(self.lib.emmax_model_finalize(self._model, self.arena.data_ptr(), nbytes, stream), "emmax_model_finalize")
        self.finalize_s = time.perf_counter() - t0     # re-layout of the bound tensors into the arena (synchronous)
        self.aux_build_s = 0.0
        del keep
        self.aux_arena: Optional[torch.Tensor] = None    # the batch >= 3 weight copies, built when a batch >= 3 first decodes
        self.max_batch = self.max_prompt = self.max_ctx = 0
        self.new_session(max_batch, max_prompt, max_ctx)

    # ------------------------------------------------------------------------------------------------------------------
    def new_session(self, max_batch: int, max_prompt: int, max_ctx: Optional[int] = None, stage_rows: int = 0) -> None:
        """(Re)create the session.  `stage_rows` staging rows (overlapped slot admissions, include/emmax.h) cost per-row state and
        their share of the paged KV region: only slot serving asks for them (ensure_stage_rows)."""
        if self._session:
            self.lib.emmax_session_destroy(self._session)
            self._session = C.c_void_p()
        np_ = self.cfg.n_patches
        if max_ctx is None:
            max_ctx = np_ + max_prompt + 512 + 1
        ws, kv = C.c_int64(), C.c_int64()
        with _lib.tuning(exact=int(self.exact)):   # the session kind (and the KV format) is read when the session is sized and created
            _lib.check(self.lib.emmax_session_bytes_ex(self._model, max_batch, max_prompt, max_ctx, int(stage_rows), C.byref(ws), C.byref(kv)),
                       "emmax_session_bytes_ex")
            self.workspace = torch.empty(ws.value, dtype=torch.uint8, device=self.device)
            self.kv = _device_bytes(kv.value, self.device, "KV")
            _lib.check(self.lib.emmax_session_create_ex(self._model, max_batch, max_prompt, max_ctx, int(stage_rows), self.workspace.data_ptr(),
                                                        ws.value, self.kv.data_ptr(), kv.value, C.byref(self._session)),
                       "emmax_session_create_ex")
        assert bool(self.lib.emmax_session_exact(self._session)) == bool(self.exact)
        self.max_batch, self.max_prompt, self.max_ctx, self.stage_rows = max_batch, max_prompt, max_ctx, int(stage_rows)

    def ensure_stage_rows(self, n: int) -> None:
        """Slot serving with overlapped admission stages up to `n` requests at a time: re-create the session with that many staging
        rows if it has fewer (drops any state of the current session -- call before slots_open)."""
        if n > getattr(self, "stage_rows", 0):
            # staging rows are bounded by the decode batch: grow that first, so that n_slots > max_batch surfaces as what it is
            self.new_session(max(self.max_batch, int(n)), self.max_prompt, self.max_ctx, stage_rows=int(n))

    def ensure_decode_batch(self, batch: int) -> None:
        """Decode batches >= 3 read MFMA-fragment-major weight copies that live in a second arena (include/emmax.h:
        emmax_model_build_aux); a model that only ever serves batches 1-2 never pays for them (13 GB at 7B)."""
        if batch < 3 or self.aux_arena is not None:
            return
        n = int(self.lib.emmax_model_aux_bytes(self._model))
        import time

        t0 = time.perf_counter()
        self.aux_arena = _device_bytes(max(n, 256), self.device, "AUX")
        _lib.check(self.lib.emmax_model_build_aux(self._model, self.aux_arena.data_ptr(), n, _lib.current_stream()), "emmax_model_build_aux")
        self.aux_build_s = time.perf_counter() - t0

    @property
    def patch_dtype(self) -> torch.dtype:
        """Element type of the patch embeddings the session hands out and takes back: bf16, fp32 in exact numerics (include/emmax.h)."""
        return torch.float32 if self.exact else torch.bfloat16

    def max_decode_batch(self) -> int:
        """Rows of one decode batch this model can run (16 for LLaMA-2-7B shapes, 8 for shapes outside decode_km.hip)."""
        if self.exact:      # the two-term kernels take 8 rows per launch and larger batches run in chunks of 8, whatever the default kernels' shape limits
            return 64
        return int(self.lib.emmax_model_max_decode_batch(self._model))

    def weight_bytes(self) -> int:
        return int(self.arena.numel()) + (int(self.aux_arena.numel()) if self.aux_arena is not None else 0)

    def ensure_capacity(self, batch: int, prompt: int, max_new: int) -> None:
        need_ctx = self.cfg.n_patches + prompt + max_new + 1
        if batch > self.max_batch or prompt > self.max_prompt or need_ctx > self.max_ctx:
            self.new_session(max(batch, self.max_batch), max(prompt, self.max_prompt), max(need_ctx, self.max_ctx),
                             stage_rows=getattr(self, "stage_rows", 0))

    def close(self) -> None:
        if getattr(self, "_session", None):
            self.lib.emmax_session_destroy(self._session)
            self._session = C.c_void_p()
        if getattr(self, "_model", None):
            self.lib.emmax_model_destroy(self._model)
            self._model = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------------------------
    def vision_encode(self, frames_u8: torch.Tensor) -> torch.Tensor:
        """uint8 [B,224,224,3] on device -> [B,256,hidden] projected patch embeddings (bf16; fp32 in an exact-numerics session: `patch_dtype`)."""
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_cuda and frames_u8.is_contiguous()
        frames_u8 = self.resize_frames(frames_u8)   # no-op at the native 224x224
        B = frames_u8.shape[0]
        out = torch.empty(B, self.cfg.n_patches, self.cfg.llm.hidden_size, dtype=self.patch_dtype, device=self.device)
        _lib.check(self.lib.emmax_vision_encode(self._session, frames_u8.data_ptr(), B, out.data_ptr(), _lib.current_stream()),
                   "emmax_vision_encode")
        return out

    def resize_frames(self, frames_u8: torch.Tensor, size: Optional[int] = None) -> torch.Tensor:
        """uint8 [B,H,W,3] on device -> uint8 [B,size,size,3], bit-exact with PIL.Image.resize(BICUBIC) (resize-naive)."""
        from .resize import bicubic_coeffs

        size = size or self.cfg.towers[0].image_size
        B, H, W, _ = frames_u8.shape
        if H == size and W == size:
            return frames_u8
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_cuda and frames_u8.is_contiguous()
        key = (H, W, size)
        if not hasattr(self, "_resize_tables"):
            self._resize_tables = {}
        if key not in self._resize_tables:
            bh, kh, nh = bicubic_coeffs(W, size)
            bv, kv, nv = bicubic_coeffs(H, size)
            self._resize_tables[key] = tuple(torch.from_numpy(a.copy()).to(self.device) for a in (bh, kh, bv, kv)) + (nh, nv)
        bh, kh, bv, kv, nh, nv = self._resize_tables[key]
        out = torch.empty(B, size, size, 3, dtype=torch.uint8, device=self.device)
        tmp = torch.empty(B, H, size, 3, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.emmax_op_resize_bicubic_u8(frames_u8.data_ptr(), B, H, W, out.data_ptr(), size, size, tmp.data_ptr(),
                                                       bh.data_ptr(), kh.data_ptr(), nh, bv.data_ptr(), kv.data_ptr(), nv,
                                                       _lib.current_stream()), "emmax_op_resize_bicubic_u8")
        return out

    def vision_encode_pixels(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """bf16 [B,6,224,224] (PrismaticProcessor layout) -> bf16 [B,256,hidden]."""
        pv = pixel_values.to(device=self.device, dtype=torch.bfloat16).contiguous()
        B = pv.shape[0]
        out = torch.empty(B, self.cfg.n_patches, self.cfg.llm.hidden_size, dtype=self.patch_dtype, device=self.device)
        _lib.check(self.lib.emmax_vision_encode_pixels(self._session, pv.data_ptr(), B, out.data_ptr(), _lib.current_stream()),
                   "emmax_vision_encode_pixels")
        return out

    def vision_features(self, B: int) -> torch.Tensor:
        out = torch.empty(B, self.cfg.n_patches, self.cfg.vision_dim, dtype=torch.bfloat16, device=self.device)
        _lib.check(self.lib.emmax_vision_features(self._session, B, out.data_ptr(), _lib.current_stream()), "emmax_vision_features")
        return out

    def prefill(self, input_ids: Sequence[Sequence[int]], patch_embeds: torch.Tensor) -> List[int]:
        """Ragged prompts (row b = list of ids starting with BOS). Returns per-row packed lengths S_b = 256 + P_b."""
        B = len(input_ids)
        self.ensure_decode_batch(B)
        lens = [len(r) for r in input_ids]
        P_max = max(lens)
        ids = torch.full((B, P_max), self.cfg.pad_token_id, dtype=torch.int32)
        for b, r in enumerate(input_ids):
            ids[b, : len(r)] = torch.as_tensor(list(r), dtype=torch.int32)
        ids_d = ids.to(self.device)
        lens_c = (C.c_int32 * B)(*lens)
        if patch_embeds is None:   # language-only forward
            _lib.check(self.lib.emmax_prefill_text(self._session, ids_d.data_ptr(), lens_c, B, P_max, _lib.current_stream()),
                       "emmax_prefill_text")
            self._last_S = list(lens)
            self._last_B = B
            return self._last_S
        pe = patch_embeds.contiguous()
        assert pe.dtype == self.patch_dtype and pe.shape[0] == B
        _lib.check(self.lib.emmax_prefill(self._session, ids_d.data_ptr(), lens_c, B, P_max, pe.data_ptr(), _lib.current_stream()),
                   "emmax_prefill")
        self._last_S = [self.cfg.n_patches + n for n in lens]
        self._last_B = B
        return self._last_S

    def prefill_logits(self) -> List[torch.Tensor]:
        """f32 logits of every prefill position, one [S_b, vocab] tensor per row."""
        total = sum(self._last_S)
        out = torch.empty(total, self.cfg.llm.vocab_size, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.emmax_prefill_logits(self._session, out.data_ptr(), _lib.current_stream()), "emmax_prefill_logits")
        return list(torch.split(out, self._last_S, dim=0))

    def last_logits(self) -> torch.Tensor:
        out = torch.empty(self._last_B, self.cfg.llm.vocab_size, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.emmax_last_logits(self._session, out.data_ptr(), _lib.current_stream()), "emmax_last_logits")
        return out

    def decode_step(self) -> None:
        _lib.check(self.lib.emmax_decode_step(self._session, _lib.current_stream()), "emmax_decode_step")

    def set_current_tokens(self, toks: Sequence[int]) -> None:
        t = torch.as_tensor(list(toks), dtype=torch.int32).to(self.device)
        _lib.check(self.lib.emmax_set_current_tokens(self._session, t.data_ptr(), _lib.current_stream()), "emmax_set_current_tokens")
        torch.cuda.current_stream().synchronize()   # `t` must outlive the copy

    def generate(self, max_new_tokens: int, stop_on_eos: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        """Greedy loop after prefill. Returns (ids int32 [B,max_new] padded with pad_id, lens int32 [B]) on device."""
        B = self._last_B
        out = torch.empty(B, max_new_tokens, dtype=torch.int32, device=self.device)
        lens = torch.empty(B, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.emmax_generate(self._session, max_new_tokens, int(stop_on_eos), out.data_ptr(), lens.data_ptr(),
                                           _lib.current_stream()), "emmax_generate")
        return out, lens

    # ---- early exit + slot serving (include/emmax.h "slot serving") ----------------------------------------------------
    def set_stop(self, trigger_ids: Sequence[int] = (), n_after: int = 0) -> None:
        """Device-side early exit: a row is done `n_after` tokens after it emitted `trigger_ids`; () clears the rule."""
        trig = list(trigger_ids)
        arr = (C.c_int32 * max(1, len(trig)))(*trig)
        _lib.check(self.lib.emmax_session_set_stop(self._session, arr, len(trig), int(n_after), _lib.current_stream()),
                   "emmax_session_set_stop")

    def slots_open(self, n_slots: int) -> None:
        self.ensure_decode_batch(int(n_slots))
        _lib.check(self.lib.emmax_slots_open(self._session, int(n_slots), _lib.current_stream()), "emmax_slots_open")
        self._n_slots = int(n_slots)
        self._slot_state = torch.empty(2, n_slots, dtype=torch.int32, device=self.device)

    def slot_prefill(self, slot: int, input_ids: Sequence[int], patch_embeds: Optional[torch.Tensor], max_new_tokens: int) -> None:
        """Prefill one request (prompt ids starting with BOS, its [n_patches, hidden] bf16 patch embeddings) into `slot`."""
        ids_d = torch.as_tensor(list(input_ids), dtype=torch.int32).to(self.device)
        pe = None
        if patch_embeds is not None:
            pe = patch_embeds.contiguous()
            assert pe.dtype == self.patch_dtype and pe.numel() == self.cfg.n_patches * self.cfg.llm.hidden_size
        _lib.check(self.lib.emmax_slot_prefill(self._session, int(slot), ids_d.data_ptr(), len(input_ids), _lib.ptr(pe),
                                               int(max_new_tokens), _lib.current_stream()), "emmax_slot_prefill")
        torch.cuda.current_stream().synchronize()   # `ids_d` must outlive the embedding gather

    def slots_prefill(self, slot0: int, prompts: Sequence[Sequence[int]], patch_embeds: Optional[Sequence[torch.Tensor]],
                      max_new_tokens: Sequence[int]) -> None:
        """Prefill len(prompts) requests into the consecutive slots slot0.. in ONE packed pass (ragged prompt lengths)."""
        n = len(prompts)
        if n < 1 or len(max_new_tokens) != n:
            raise ValueError("slots_prefill: one token budget per prompt, at least one prompt")
        lens = [len(p) for p in prompts]
        P = max(lens)
        ids = torch.zeros(n, P, dtype=torch.int32)
        for i, p in enumerate(prompts):
            ids[i, : lens[i]] = torch.as_tensor(list(p), dtype=torch.int32)
        ids_d = ids.to(self.device)
        pe = None
        if patch_embeds is not None:
            if len(patch_embeds) != n:
                raise ValueError("slots_prefill: one patch-embedding tensor per prompt")
            pe = torch.stack([t.reshape(self.cfg.n_patches, self.cfg.llm.hidden_size) for t in patch_embeds]).contiguous()
            assert pe.dtype == self.patch_dtype
        lens_c = (C.c_int32 * n)(*lens)
        budget_c = (C.c_int32 * n)(*[int(v) for v in max_new_tokens])
        _lib.check(self.lib.emmax_slots_prefill(self._session, int(slot0), n, ids_d.data_ptr(), P, lens_c, _lib.ptr(pe), budget_c,
                                                _lib.current_stream()), "emmax_slots_prefill")
        torch.cuda.current_stream().synchronize()   # `ids_d` / `pe` must outlive the pass

    # ---- overlapped admission: staged prefill on a second stream + commit between two decode steps (include/emmax.h) ----
    def admission(self):
        """Context manager: everything issued inside (frame encode, staged prefill, their uploads) runs on the engine's admission
        stream, beside the decode steps of the current stream; ordered after the previous commit."""
        if getattr(self, "_adm_stream", None) is None:
            self._adm_stream = torch.cuda.Stream(device=self.device)
            self._commit_event = None
        if self._commit_event is not None:
            self._adm_stream.wait_event(self._commit_event)
        return torch.cuda.stream(self._adm_stream)

    def slots_prefill_staged(self, prompts: Sequence[Sequence[int]], patch_embeds: Optional[Sequence[torch.Tensor]],
                             max_new_tokens: Sequence[int]):
        """Prefill len(prompts) requests into the session's STAGING rows; call inside `with engine.admission():`.  Returns a handle
        (`.ready()` -> bool, non-blocking) for `slots_commit`."""
        n = len(prompts)
        if n < 1 or len(max_new_tokens) != n:
            raise ValueError("slots_prefill_staged: one token budget per prompt, at least one prompt")
        lens = [len(p) for p in prompts]
        P = max(lens)
        ids = torch.zeros(n, P, dtype=torch.int32)
        for i, p in enumerate(prompts):
            ids[i, : lens[i]] = torch.as_tensor(list(p), dtype=torch.int32)
        ids_d = ids.to(self.device)
        pe = None
        if patch_embeds is not None:
            if len(patch_embeds) != n:
                raise ValueError("slots_prefill_staged: one patch-embedding tensor per prompt")
            pe = torch.stack([t.reshape(self.cfg.n_patches, self.cfg.llm.hidden_size) for t in patch_embeds]).contiguous()
            assert pe.dtype == self.patch_dtype
        lens_c = (C.c_int32 * n)(*lens)
        budget_c = (C.c_int32 * n)(*[int(v) for v in max_new_tokens])
        _lib.check(self.lib.emmax_slots_prefill_staged(self._session, n, ids_d.data_ptr(), P, lens_c, _lib.ptr(pe), budget_c,
                                                       _lib.current_stream()), "emmax_slots_prefill_staged")
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return _Staged(n, ev, (ids_d, pe))

    def slots_commit(self, staged: "_Staged", slots: Sequence[int], staged_idx: Optional[Sequence[int]] = None) -> None:
        """Move staged requests `staged_idx` (default: the first len(slots) not yet committed) into `slots` (idle) on the CURRENT
        (decode) stream, after the staged prefill has finished.  A batch may be committed piecemeal as slots free up."""
        k = len(slots)
        if staged_idx is None:
            staged_idx = list(range(staged.committed, staged.committed + k))
        if k < 1 or len(staged_idx) != k or max(staged_idx) >= staged.n:
            raise ValueError("slots_commit: one staged request per slot")
        torch.cuda.current_stream().wait_event(staged.event)
        sl = (C.c_int32 * k)(*[int(v) for v in slots])
        si = (C.c_int32 * k)(*[int(v) for v in staged_idx])
        _lib.check(self.lib.emmax_slots_commit(self._session, si, sl, k, _lib.current_stream()), "emmax_slots_commit")
        staged.committed += k
        self._commit_event = torch.cuda.Event()
        self._commit_event.record(torch.cuda.current_stream())
        if staged.committed >= staged.n:
            staged.keep = None

    def slots_step(self, n_steps: int) -> None:
        _lib.check(self.lib.emmax_slots_step(self._session, int(n_steps), _lib.current_stream()), "emmax_slots_step")

    def slots_state(self) -> Tuple[List[int], List[int]]:
        """(done flags, generated-token counts) of every slot; synchronises with the device."""
        st = self._slot_state
        _lib.check(self.lib.emmax_slots_state(self._session, st[0].data_ptr(), st[1].data_ptr(), _lib.current_stream()),
                   "emmax_slots_state")
        host = st.cpu()
        return host[0].tolist(), host[1].tolist()

    def slot_output(self, slot: int, n: int) -> List[int]:
        out = torch.empty(max(1, n), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.emmax_slot_output(self._session, int(slot), out.data_ptr(), int(n), _lib.current_stream()),
                   "emmax_slot_output")
        return out[:n].cpu().tolist()

    def slot_release(self, slot: int) -> None:
        _lib.check(self.lib.emmax_slot_release(self._session, int(slot), _lib.current_stream()), "emmax_slot_release")

    def graph_active(self) -> bool:
        return bool(self.lib.emmax_session_graph_active(self._session))

    def profile_decode_stage(self, stage: int, reps: int = 3) -> float:
        """Mean duration (microseconds) of one launch of decode stage `stage` (see include/emmax.h), HIP-event timed."""
        us = C.c_float()
        _lib.check(self.lib.emmax_profile_decode_stage(self._session, stage, reps, C.byref(us), _lib.current_stream()),
                   "emmax_profile_decode_stage")
        return float(us.value)
