"""Slot scheduler host logic (emmax/serving.py) against a fake engine: no GPU, no HIP library.

The fake engine mimics the device-side contract of the slot C ABI (include/emmax.h "slot serving"): a slot emits one
token per decode step until its planned length, its budget or its stop rule is reached; finished / idle slots stay put."""
import pytest

from emmax.serving import Request, SlotScheduler, serve_static


class FakeEngine:
    def __init__(self, plans):
        self.plans = plans            # rid-keyed: list of ids the "model" would emit to EOS
        self.n = 0
        self.slots = {}
        self.prefills = []
        self.steps = 0
        self.stop = ([], 0)
        self.released = []

    def set_stop(self, trig, n_after):
        self.stop = (list(trig), n_after)

    def slots_open(self, n):
        self.n = n
        self.slots = {s: None for s in range(n)}

    def _done(self, st):
        if len(st["out"]) >= min(len(st["plan"]), st["budget"]):
            return True
        trig, after = self.stop
        if trig:
            out = st["out"]
            for i in range(len(out) - len(trig) + 1):
                if out[i:i + len(trig)] == trig:
                    return len(out) - (i + len(trig)) >= after
        return False

    def slot_prefill(self, slot, ids, pe, max_new):
        assert self.slots[slot] is None, "prefill into a busy slot"
        rid = pe["rid"]
        assert pe["encoded"] is True
        st = {"plan": self.plans[rid], "budget": max_new, "out": [self.plans[rid][0]], "rid": rid}
        self.slots[slot] = st
        self.prefills.append((slot, rid, tuple(ids)))

    def slots_step(self, k):
        self.steps += k
        for _ in range(k):
            for st in self.slots.values():
                if st is not None and not self._done(st):
                    st["out"].append(st["plan"][len(st["out"])])

    def slots_state(self):
        done = [1 if (st is None or self._done(st)) else 0 for st in self.slots.values()]
        n_out = [0 if st is None else len(st["out"]) for st in self.slots.values()]
        return done, n_out

    def slot_output(self, slot, n):
        return list(self.slots[slot]["out"][:n])

    def slot_release(self, slot):
        self.released.append(slot)
        self.slots[slot] = None


def _encode_factory(calls):
    def encode(frames):
        calls.append(len(frames))
        return [{"rid": f, "encoded": True} for f in frames]
    return encode


def _plans(lengths):
    return {i: [100 * i + t for t in range(n)] for i, n in enumerate(lengths)}


def test_every_request_completes_with_its_own_output():
    lengths = [5, 40, 9, 17, 3, 64, 21, 8, 33, 12, 6]
    plans = _plans(lengths)
    eng, calls = FakeEngine(plans), []
    sch = SlotScheduler(eng, _encode_factory(calls), n_slots=3, poll_every=4)
    for i in range(len(lengths)):
        sch.submit(Request(rid=i, frame=i, prompt_ids=[1, 7, i + 3], max_new_tokens=512))
    res = sch.run()
    assert sorted(r.rid for r in res) == list(range(len(lengths)))
    for r in res:
        assert r.ids == plans[r.rid]
        assert r.t_submit <= r.t_admit <= r.t_done
    assert calls[0] == 3 and sum(calls) == len(lengths)        # the first admission round encodes a full batch of frames
    assert len(eng.released) == len(lengths)
    assert all(st is None for st in eng.slots.values())
    # a slot is reused as soon as it frees up: far fewer steps than static batching (sum of per-batch maxima)
    static_steps = sum(max(lengths[i:i + 3]) for i in range(0, len(lengths), 3))
    assert sch.steps < static_steps + 3 * 4


def test_budget_and_stop_rule_end_a_request_early():
    plans = {0: list(range(10, 60)), 1: [5, 6, 7, 29871, 31800, 31801, 31802, 31803, 31804, 31805, 31806, 2, 9, 9]}
    eng = FakeEngine(plans)
    sch = SlotScheduler(eng, _encode_factory([]), n_slots=2, poll_every=1, stop_trigger=[29871], stop_after=7)
    assert eng.stop == ([29871], 7)
    sch.submit(Request(0, 0, [1, 2], max_new_tokens=12))
    sch.submit(Request(1, 1, [1, 3], max_new_tokens=512))
    res = {r.rid: r for r in sch.run()}
    assert res[0].ids == plans[0][:12]                          # token budget
    assert res[1].ids == plans[1][:11]                          # trigger + 7 tokens, EOS never decoded


def test_completion_order_and_slot_reuse():
    plans = _plans([30, 2, 2, 2])
    eng = FakeEngine(plans)
    sch = SlotScheduler(eng, _encode_factory([]), n_slots=2, poll_every=2)
    for i in range(4):
        sch.submit(Request(i, i, [1, i + 5]))
    res = sch.run()
    assert [r.rid for r in res] == [1, 2, 3, 0]                 # the long request never blocks the short ones
    assert {slot for slot, rid, _ in eng.prefills if rid in (1, 2, 3)} == {1}   # they all went through the same slot
    assert [r.slot for r in res] == [1, 1, 1, 0]


def test_encode_ahead_batches_the_vit_and_keeps_results():
    lengths = [9, 3, 12, 5, 7, 4, 10, 6, 8, 2]
    plans = _plans(lengths)
    eng, calls = FakeEngine(plans), []
    sch = SlotScheduler(eng, _encode_factory(calls), n_slots=2, poll_every=1, encode_ahead=4)
    for i in range(len(lengths)):
        sch.submit(Request(i, i, [1, 4, i + 3]))
    res = sch.run()
    assert {r.rid: r.ids for r in res} == plans
    assert sum(calls) == len(lengths)                  # every frame is encoded exactly once ...
    assert max(calls) == 4 and len(calls) <= 4         # ... in batches of up to 4 instead of one call per admission
    assert sch._embeds == {}


def test_bad_arguments():
    eng = FakeEngine({})
    with pytest.raises(ValueError):
        SlotScheduler(eng, lambda f: f, n_slots=65)   # 64 = the decode step's row limit (round 6; round 5: 32; rounds 1-4: 8)
    with pytest.raises(ValueError):
        SlotScheduler(eng, lambda f: f, n_slots=2, poll_every=0)
    sch = SlotScheduler(eng, lambda f: f, n_slots=2)
    with pytest.raises(ValueError):
        sch.submit(Request(0, 0, []))
    assert sch.run() == []


def test_encode_count_mismatch_is_an_error():
    eng = FakeEngine(_plans([3]))
    sch = SlotScheduler(eng, lambda frames: [], n_slots=1)
    sch.submit(Request(0, 0, [1, 2]))
    with pytest.raises(RuntimeError):
        sch.run()


def test_static_baseline_helper():
    reqs = [Request(i, i, [1]) for i in range(5)]
    seen = []

    def gen(batch):
        seen.append(len(batch))
        return [[r.rid] for r in batch]

    assert serve_static(gen, reqs, 2) == [[0], [1], [2], [3], [4]]
    assert seen == [2, 2, 1]


class BatchingFakeEngine(FakeEngine):
    """The fake engine with the packed multi-slot prefill of the C ABI (emmax_slots_prefill)."""

    def __init__(self, plans):
        super().__init__(plans)
        self.packed = []

    def slots_prefill(self, slot0, prompts, embeds, budgets):
        assert len(prompts) == len(embeds) == len(budgets) >= 2
        self.packed.append((slot0, len(prompts)))
        for i, (ids, pe, mx) in enumerate(zip(prompts, embeds, budgets)):
            self.slot_prefill(slot0 + i, ids, pe, mx)


def test_consecutive_free_slots_are_prefilled_in_one_packed_pass():
    lengths = [30, 4, 4, 30, 12, 12, 9, 9, 9]
    plans = _plans(lengths)
    eng = BatchingFakeEngine(plans)
    sch = SlotScheduler(eng, _encode_factory([]), n_slots=4, poll_every=4)
    for i in range(len(lengths)):
        sch.submit(Request(rid=i, frame=i, prompt_ids=[1, 7, i + 3], max_new_tokens=512))
    res = sch.run()
    assert sorted(r.rid for r in res) == list(range(len(lengths)))
    for r in res:
        assert r.ids == plans[r.rid]
    assert eng.packed[0] == (0, 4)                    # the initial fill: all four slots in one pass
    assert (1, 2) in eng.packed                       # requests 1 and 2 finish together: slots 1, 2 refilled as one run
    # slots 0 and 3 (the long requests) free up together later but are NOT neighbours: two single-row prefills
    singles = [slot for slot, rid, _ in eng.prefills if rid in (6, 7, 8)]
    assert all((s0, n) != (0, 2) for s0, n in eng.packed) and len(singles) == 3
    # every request went through exactly one prefill
    assert sorted(rid for _, rid, _ in eng.prefills) == list(range(len(lengths)))


class FakeStagedEngine(FakeEngine):
    """FakeEngine + the overlapped-admission contract (engine.admission / slots_prefill_staged / slots_commit): a staged prefill
    becomes ready only after `lag` further decode steps, and touches no slot until it is committed."""

    def __init__(self, plans, lag=3):
        super().__init__(plans)
        self.lag, self.in_admission, self.staged, self.commits = lag, False, None, []

    def admission(self):
        eng = self

        class Ctx:
            def __enter__(self):
                eng.in_admission = True

            def __exit__(self, *a):
                eng.in_admission = False

        return Ctx()

    def slots_prefill_staged(self, prompts, embeds, max_new):
        assert self.in_admission and self.staged is None, "one staged batch at a time, issued on the admission stream"
        eng, due = self, self.steps + self.lag

        class H:
            n = len(prompts)

            def ready(self):
                return eng.steps >= due

            def wait(self):
                eng.steps = max(eng.steps, due)      # nothing else was running: the host blocks

        self.staged = (H(), [{"plan": self.plans[pe["rid"]], "budget": b, "out": [self.plans[pe["rid"]][0]], "rid": pe["rid"]}
                              for pe, b in zip(embeds, max_new)], [tuple(p) for p in prompts])
        assert all(pe["encoded"] for pe in embeds)
        return self.staged[0]

    def slots_commit(self, handle, slots, staged_idx=None):
        h, sts, prompts = self.staged
        assert h is handle and h.ready() and not self.in_admission and len(slots) == len(staged_idx) >= 1
        for slot, i in zip(slots, staged_idx):
            assert self.slots[slot] is None, "commit into a busy slot"
            assert sts[i] is not None, "staged request committed twice"
            self.slots[slot] = sts[i]
            self.prefills.append((slot, sts[i]["rid"], prompts[i]))
            sts[i] = None
        self.commits.append(list(slots))
        if all(st is None for st in sts):
            self.staged = None


def test_overlapped_admission_keeps_decoding_while_a_request_is_prefilled():
    lengths = [30, 4, 4, 9, 25, 3, 14, 6, 40, 5]
    plans = _plans(lengths)
    eng, calls = FakeStagedEngine(plans, lag=3), []
    sch = SlotScheduler(eng, _encode_factory(calls), n_slots=3, poll_every=2, encode_ahead=2)
    assert sch.overlap
    for i in range(len(lengths)):
        sch.submit(Request(rid=i, frame=i, prompt_ids=[1, 9, i + 3]))
    res = sch.run()
    assert sorted(r.rid for r in res) == list(range(len(lengths)))
    for r in res:
        assert r.ids == plans[r.rid] and r.t_submit <= r.t_admit <= r.t_done
    assert sch.overlapped_admissions == len(eng.commits) >= 4 and eng.staged is None and sch.stage_batch == 1
    assert sum(len(c) for c in eng.commits) == len(lengths)
    # the same requests through the blocking admission give the same outputs
    eng2 = FakeEngine(plans)
    sch2 = SlotScheduler(eng2, _encode_factory([]), n_slots=3, poll_every=2)
    assert not sch2.overlap
    for i in range(len(lengths)):
        sch2.submit(Request(rid=i, frame=i, prompt_ids=[1, 9, i + 3]))
    assert {r.rid: r.ids for r in sch2.run()} == {r.rid: r.ids for r in res}
    with pytest.raises(ValueError):
        SlotScheduler(FakeEngine(plans), _encode_factory([]), n_slots=2, overlap=True)


def test_a_failed_staged_prefill_requeues_its_requests_and_asks_for_staging_rows():
    """`_start_admission` pops its batch off the queue before the engine call; if that call raises, the requests used to
    be neither requeued nor reported.  Now they go back to the head of the queue in their order and the error reaches the caller --
    a retry then serves everything.  Round 5 also: an overlapped scheduler asks the engine for as many staging rows as it has slots
    (sessions hold none unless asked)."""
    lengths = [5, 7, 3, 9, 4, 6]
    plans = _plans(lengths)

    class Flaky(FakeStagedEngine):
        fail_next, asked = True, None

        def ensure_stage_rows(self, n):
            self.asked = n

        def slots_prefill_staged(self, prompts, embeds, max_new):
            if self.fail_next:
                self.fail_next = False
                raise RuntimeError("staged prefill failed")
            return super().slots_prefill_staged(prompts, embeds, max_new)

    eng = Flaky(plans, lag=1)
    sch = SlotScheduler(eng, _encode_factory([]), n_slots=3, poll_every=1)
    assert sch.overlap and eng.asked == 3
    for i in range(len(lengths)):
        sch.submit(Request(rid=i, frame=i, prompt_ids=[1, 9, i + 3]))
    with pytest.raises(RuntimeError, match="staged prefill failed"):
        sch.run()
    assert [r.rid for r, _ in sch.queue] == list(range(len(lengths))) and sch._pending is None and not sch.active   # nothing lost, order kept
    res = sch.run()                                                                                                   # the retry
    assert sorted(r.rid for r in res) == list(range(len(lengths))) and all(r.ids == plans[r.rid] for r in res)
