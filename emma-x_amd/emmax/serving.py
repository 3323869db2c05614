"""Continuous batching for a fleet of cameras: a slot scheduler over the engine's slot-serving C ABI.

The reference answers one request at a time (`generate_actions`, prismatic.py:634-696: bs = 1, always decoded to EOS or
512 new tokens).  Here the decode batch is a set of independent request SLOTS: every slot owns its KV pages, context length,
token budget and stop state on the device; finished slots are refilled (vision encode + prefill, consecutive free slots in one packed pass) while the other
slots keep decoding, so a long reasoning chain never holds short ones back.  With a stop rule (`stop_trigger`,
`stop_after`) a request ends as soon as its action line is complete instead of at EOS -- the emitted ids are exactly the
prefix of the full greedy generation, so the parsed action is unchanged.

There is no shareable prompt prefix to cache across requests: the spliced sequence is [BOS] + 256 patch embeddings +
text[1:] (modeling_prismatic.py:379-384), so everything after BOS depends on the frame.

The scheduler is host logic only (testable without a GPU against a fake engine); all state it polls lives on the device.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Any, Callable, Deque, Dict, List, Optional, Sequence
from collections import deque


@dataclass
class Request:
    """One camera frame + prompt.  `frame` is whatever `encode` accepts (uint8 [H,W,3] tensor on the device)."""
    rid: Any
    frame: Any
    prompt_ids: Sequence[int]
    max_new_tokens: int = 512


@dataclass
class Result:
    rid: Any
    ids: List[int]
    slot: int
    t_submit: float
    t_admit: float
    t_done: float

    @property
    def latency_s(self) -> float:
        return self.t_done - self.t_submit


@dataclass
class _Active:
    req: Request
    t_submit: float
    t_admit: float


MAX_SLOTS = 64   # EMMAX_MAX_DECODE_BATCH (emma-x_amd/csrc/kernels.h): rows of a decode step


class SlotScheduler:
    """Greedy slot scheduler.

    engine   object with slots_open / slot_prefill / slots_step / slots_state / slot_output / slot_release / set_stop
             (emmax.engine.EmmaxEngine, or a fake in the CPU tests)
    encode   frames (list) -> patch embeddings, one [n_patches, hidden] bf16 tensor per frame; called once per admission
             round with every frame admitted in that round, so the ViT runs batched
    n_slots  decode batch (<= 64; the engine's max_decode_batch())
    poll_every   decode steps between two device polls (a poll is one tiny D2H copy + sync; default 4 = ~13 ms at 7B shapes: a retired
                 slot is refilled within 4 steps -- 32 requests / 8 slots: 22.9 actions/s polling every 16 steps, 24.2 every 4)
    encode_ahead frames encoded per `encode` call: the frames being admitted plus the next ones in the queue, so the ViT
                 always runs at a useful batch (a lone frame costs ~6 ms, eight cost ~8 ms); their patch embeddings wait in
                 `self._embeds` (16 MB for eight 7B-shape frames) until their request is admitted
    """

    def __init__(self, engine, encode: Callable[[List[Any]], List[Any]], n_slots: int = 8, poll_every: int = 4,
                 stop_trigger: Sequence[int] = (), stop_after: int = 0, clock: Callable[[], float] = time.perf_counter,
                 encode_ahead: int = 0, overlap: Optional[bool] = None, stage_batch: Optional[int] = None) -> None:
        if not 1 <= n_slots <= MAX_SLOTS:
            raise ValueError(f"n_slots {n_slots} outside 1..{MAX_SLOTS}")
        if poll_every < 1:
            raise ValueError("poll_every must be >= 1")
        self.engine = engine
        self.encode = encode
        self.n_slots = n_slots
        self.poll_every = poll_every
        self.clock = clock
        self.queue: Deque[tuple] = deque()
        self.active: Dict[int, _Active] = {}
        self.results: List[Result] = []
        self.steps = 0
        self.polls = 0
        self.encode_ahead = max(int(encode_ahead), 0)
        self._embeds: Dict[int, Any] = {}          # id(request) -> patch embeddings encoded ahead of admission
        # overlap: admissions (frame encode + prefill) run on the engine's admission stream into staging rows while the occupied
        # slots keep decoding, and join their slots between two decode steps (engine.admission / slots_prefill_staged /
        # slots_commit).  Default: on whenever the engine has the staged API.
        has = all(hasattr(engine, a) for a in ("admission", "slots_prefill_staged", "slots_commit"))
        if overlap and not has:
            raise ValueError("overlap=True needs an engine with admission / slots_prefill_staged / slots_commit")
        self.overlap = has if overlap is None else bool(overlap)
        # requests are staged AHEAD of need, `stage_batch` at a time (default: half the slots), whether or not a slot is free: a slot
        # that retires finds its successor prefilled and is refilled at the same poll
        self.stage_batch = max(1, min(int(stage_batch) if stage_batch else max(1, n_slots // 2), n_slots))
        self._pending = None                         # [staged handle, [(request, t_submit), ...], committed so far] of the staged batch
        self.overlapped_admissions = 0
        # a staged batch is at most max(stage_batch, free slots) <= n_slots requests: the session needs that many staging rows
        # (they cost KV pages, so a session only has them when a scheduler asks: engine.ensure_stage_rows re-creates it if needed)
        if self.overlap and hasattr(engine, "ensure_stage_rows"):
            engine.ensure_stage_rows(n_slots)
        engine.set_stop(list(stop_trigger), stop_after)
        engine.slots_open(n_slots)

    def submit(self, req: Request) -> None:
        if len(req.prompt_ids) < 1:
            raise ValueError("empty prompt")
        self.queue.append((req, self.clock()))

    def _admit(self) -> int:
        free = [s for s in range(self.n_slots) if s not in self.active]
        take = min(len(free), len(self.queue))
        if take == 0:
            return 0
        batch = [self.queue.popleft() for _ in range(take)]
        embeds = self._encode_for(batch)
        # runs of CONSECUTIVE free slots are prefilled in one packed pass (eight one-row prefills cost ~1.6x one eight-row pass);
        # engines without `slots_prefill` (test doubles) get the requests one by one
        slots = free[:take]
        i = 0
        while i < take:
            j = i + 1
            while j < take and slots[j] == slots[j - 1] + 1:
                j += 1
            if j - i > 1 and hasattr(self.engine, "slots_prefill"):
                self.engine.slots_prefill(slots[i], [list(batch[k][0].prompt_ids) for k in range(i, j)], embeds[i:j] if embeds[i] is not None else None,
                                          [batch[k][0].max_new_tokens for k in range(i, j)])
            else:
                for k in range(i, j):
                    self.engine.slot_prefill(slots[k], list(batch[k][0].prompt_ids), embeds[k], batch[k][0].max_new_tokens)
            t_adm = self.clock()
            for k in range(i, j):
                self.active[slots[k]] = _Active(batch[k][0], batch[k][1], t_adm)
            i = j
        return take

    def _encode_for(self, batch) -> List[Any]:
        """Patch embeddings of the requests in `batch` (one ViT pass for those not encoded yet + the head of the queue)."""
        todo = [r for r, _ in batch if id(r) not in self._embeds]
        if todo:
            ahead = [r for r, _ in list(self.queue)[: max(self.encode_ahead - len(todo), 0)] if id(r) not in self._embeds]
            got = self.encode([r.frame for r in todo + ahead])
            if len(got) != len(todo) + len(ahead):
                raise RuntimeError(f"encode returned {len(got)} embeddings for {len(todo) + len(ahead)} frames")
            for r, pe in zip(todo + ahead, got):
                self._embeds[id(r)] = pe
        return [self._embeds.pop(id(r)) for r, _ in batch]

    def _start_admission(self) -> bool:
        """Overlap mode: issue frame encode + staged prefill of the next `stage_batch` queued requests on the admission stream --
        ahead of need, whether or not a slot is free; returns at once (nothing is awaited)."""
        if self._pending is not None or not self.queue:
            return False
        free = sum(1 for s in range(self.n_slots) if s not in self.active)
        take = min(len(self.queue), max(self.stage_batch, free))
        batch = [self.queue.popleft() for _ in range(take)]
        try:
            with self.engine.admission():
                embeds = self._encode_for(batch)
                staged = self.engine.slots_prefill_staged([list(r.prompt_ids) for r, _ in batch], embeds if embeds[0] is not None else None,
                                                          [r.max_new_tokens for r, _ in batch])
        except Exception:
            # nothing was staged: the requests go back to the head of the queue in their order and the error reaches the caller
            # (they were popped and then neither requeued nor reported)
            for item in reversed(batch):
                self._embeds.pop(id(item[0]), None)
                self.queue.appendleft(item)
            raise
        self._pending = [staged, batch, 0]
        return True

    def _join_admission(self, wait: bool) -> int:
        """Move staged requests into the slots that are free right now, if the staged prefill has finished (wait=True: block for it
        -- the decode batch is idle, there is nothing else to do)."""
        if self._pending is None:
            return 0
        staged, batch, k0 = self._pending
        slots = [s for s in range(self.n_slots) if s not in self.active][: len(batch) - k0]
        if not slots:
            return 0
        if not staged.ready():
            if not wait:
                return 0
            staged.wait()
        self.engine.slots_commit(staged, slots, list(range(k0, k0 + len(slots))))
        t_adm = self.clock()
        for slot, (req, t_sub) in zip(slots, batch[k0:k0 + len(slots)]):
            self.active[slot] = _Active(req, t_sub, t_adm)
        self._pending[2] = k0 + len(slots)
        if self._pending[2] >= len(batch):
            self._pending = None
        self.overlapped_admissions += 1
        return len(slots)

    def _retire(self) -> int:
        done, n_out = self.engine.slots_state()
        self.polls += 1
        n = 0
        for slot in sorted(self.active):
            if done[slot]:
                a = self.active.pop(slot)
                ids = self.engine.slot_output(slot, int(n_out[slot]))
                self.engine.slot_release(slot)
                self.results.append(Result(a.req.rid, ids, slot, a.t_submit, a.t_admit, self.clock()))
                n += 1
        return n

    def run(self) -> List[Result]:
        """Serve until the queue is empty and every slot is idle.  Returns the results in completion order."""
        if self.overlap:
            while self.queue or self.active or self._pending is not None:
                self._start_admission()
                if self._join_admission(wait=not self.active):   # (idle decode batch: the admission is all there is to wait for)
                    self._start_admission()                       # staging rows free again: the next batch goes out at once
                if not self.active:
                    continue
                self.engine.slots_step(self.poll_every)
                self.steps += self.poll_every
                self._retire()
            return self.results
        while self.queue or self.active:
            self._admit()
            if not self.active:
                continue
            self.engine.slots_step(self.poll_every)
            self.steps += self.poll_every
            self._retire()
        return self.results


def serve_static(engine_generate: Callable[[List[Request]], List[List[int]]], requests: Sequence[Request], batch: int) -> List[List[int]]:
    """The baseline the scheduler is measured against: fixed batches, each decoded until its slowest row is done."""
    out: List[List[int]] = []
    for i in range(0, len(requests), batch):
        out.extend(engine_generate(list(requests[i:i + batch])))
    return out


def stop_rule_from_tokenizer(tokenizer, marker: str = "POLICIES:", n_after: int = 8) -> tuple:
    """(trigger_ids, n_after) for `SlotScheduler(stop_trigger=..., stop_after=...)` / `engine.set_stop`.

    The Solver reads the action from the first line after `marker` (policy_parser.Solver.extract_action_policies, reference
    prismatic/vla/solver.py:107-137): a leading space token plus the 7 action-bin tokens, i.e. 8 ids.

    The marker is tokenised IN CONTEXT -- as it appears in generated text, right after a newline ("...MOVEMENT:\n..\nPOLICIES:\n"):
    a SentencePiece / LLaMA tokenizer given the bare string prepends its dummy prefix (`▁POL...`, or a lone 29871 `▁`), pieces
    that never occur after `\n` in running text, so a bare-string trigger would never fire and every request would run to EOS.
    Here "\n" + marker is encoded and everything up to and including the last id of the encoding of "\n" alone is dropped;
    the result is what the model actually emits for the marker.  At most 16 ids (the device-side matcher's capacity)."""
    def enc(text):
        ids = tokenizer(text, add_special_tokens=False)["input_ids"]
        return [int(t) for t in (ids[0] if ids and isinstance(ids[0], (list, tuple)) else ids)]

    head, full = enc("\n"), enc("\n" + marker)
    k = 0
    while k < len(head) and k < len(full) and head[k] == full[k]:
        k += 1
    ids = full[k:]
    if not 1 <= len(ids) <= 16:
        raise ValueError(f"marker {marker!r} tokenises to {len(ids)} ids in context; the stop rule takes 1..16")
    return ids, int(n_after)
