"""Slot serving + early exit on a real MI355X, through the C ABI.

Parity bar: a request served through a slot -- admitted while other slots are mid-decode, at whatever batch width the live
step has -- must emit exactly the ids of its own bs = 1 `generate` (bit-exact, planted margin-boosted weights whose answer
is also known a priori); with the stop rule the emitted ids are exactly a prefix of the full generation and the decoded
action is identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PREFIX = 29871


@pytest.fixture(scope="module")
def served(device):
    from emmax.config import EmmaXConfig
    from emmax.modeling import EmmaXForActionPrediction
    from emmax.weights import synthetic_state_dict

    cfg = EmmaXConfig.tiny()
    sd = {k: v.to(torch.bfloat16) for k, v in synthetic_state_dict(cfg, seed=5, planted=True).items()}
    model = EmmaXForActionPrediction(cfg, sd).to(device, max_batch=4, max_prompt=40)
    return cfg, model


def _requests(cfg, ks, seed=77):
    from emmax.weights import planted_start_token

    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, size=(len(ks), 224, 224, 3), dtype=np.uint8)
    rows = []
    for i, k in enumerate(ks):
        n = 6 + (i % 5)
        r = [1] + [int(x) for x in rng.integers(3, 31744, size=n - 1)]
        r[-1] = planted_start_token(cfg, k)       # the planted walk emits 29871 as its k-th id, then 8 action ids, then EOS
        rows.append(r)
    return frames, rows


def test_early_exit_is_a_prefix_and_keeps_the_action(device, served):
    from emmax.weights import planted_chain

    cfg, model = served
    eng = model.engine
    frames, rows = _requests(cfg, [9, 3])
    fr = torch.from_numpy(frames).to(device)
    full_ids, full_lens = model.generate_ids(rows, frames_u8=fr, max_new_tokens=40)
    full = [full_ids[b, : int(full_lens[b])].cpu().tolist() for b in range(2)]
    for b, k in enumerate([9, 3]):
        assert full[b] == planted_chain(cfg, rows[b][-1], 40) and full[b][-1] == cfg.eos_token_id and len(full[b]) == k + 9
    eng.set_stop([PREFIX], 8)
    try:
        ids, lens = model.generate_ids(rows, frames_u8=fr, max_new_tokens=40)
    finally:
        eng.set_stop([], 0)
    stats = model.get_action_stats(None)
    for b, k in enumerate([9, 3]):
        got = ids[b, : int(lens[b])].cpu().tolist()
        assert got == full[b][:-1] and len(got) == k + 8     # k-1 ordinary ids, 29871, 8 action ids: EOS is never decoded
        assert got[-9] == PREFIX
        a_early = model.actions_from_ids(rows[b] + got, stats)
        a_full = model.actions_from_ids(rows[b] + full[b][:-1], stats)
        assert np.array_equal(a_early, a_full)
    # the rule is cleared again: the next generate runs to EOS
    ids, lens = model.generate_ids(rows[:1], frames_u8=fr[:1], max_new_tokens=40)
    assert ids[0, : int(lens[0])].cpu().tolist() == full[0]


def test_multi_id_trigger_and_restart(device, served):
    """Two-id trigger that first half-matches: the single-restart matcher must still fire on the real occurrence."""
    from emmax.weights import planted_chain

    cfg, model = served
    frames, rows = _requests(cfg, [6])
    fr = torch.from_numpy(frames).to(device)
    chain = planted_chain(cfg, rows[0][-1], 40)
    trig = chain[3:5]
    model.engine.set_stop(trig, 2)
    try:
        ids, lens = model.generate_ids(rows, frames_u8=fr, max_new_tokens=40)
    finally:
        model.engine.set_stop([], 0)
    assert ids[0, : int(lens[0])].cpu().tolist() == chain[:7]


@pytest.mark.parametrize("n_slots,poll", [(3, 4), (1, 3), (4, 16)])
def test_slot_serving_equals_bs1_generate(device, served, n_slots, poll):
    from emmax.serving import Request, SlotScheduler
    from emmax.weights import planted_chain

    cfg, model = served
    eng = model.engine
    ks = [14, 2, 7, 25, 1, 4, 11, 3, 19]
    frames, rows = _requests(cfg, ks, seed=5 + n_slots)
    fr = torch.from_numpy(frames).to(device)
    want = []
    for i in range(len(ks)):
        ids, lens = model.generate_ids(rows[i:i + 1], frames_u8=fr[i:i + 1], max_new_tokens=48)
        want.append(ids[0, : int(lens[0])].cpu().tolist())
        assert want[-1] == planted_chain(cfg, rows[i][-1], 48)

    def encode(fs):
        pe = eng.vision_encode(torch.stack(fs))
        return [pe[i] for i in range(len(fs))]

    sch = SlotScheduler(eng, encode, n_slots=n_slots, poll_every=poll, encode_ahead=4 if n_slots == 3 else 0)
    for i in range(len(ks)):
        sch.submit(Request(i, fr[i], rows[i], max_new_tokens=48))
    res = sch.run()
    assert sorted(r.rid for r in res) == list(range(len(ks)))
    for r in res:
        assert r.ids == want[r.rid], f"request {r.rid} (slot {r.slot})"
    # with more than one slot the short requests overtake the long ones
    if n_slots > 1:
        assert [r.rid for r in res] != list(range(len(ks)))
    # budget: a request capped below its natural length stops at the cap, the rest is untouched
    sch = SlotScheduler(eng, encode, n_slots=n_slots, poll_every=poll)
    sch.submit(Request("cap", fr[0], rows[0], max_new_tokens=5))
    sch.submit(Request("free", fr[1], rows[1], max_new_tokens=48))
    res = {r.rid: r.ids for r in sch.run()}
    assert res["cap"] == want[0][:5] and res["free"] == want[1]
    # and the plain batched API still works on the same session afterwards
    ids, lens = model.generate_ids(rows[:2], frames_u8=fr[:2], max_new_tokens=48)
    assert [ids[b, : int(lens[b])].cpu().tolist() for b in range(2)] == want[:2]


def test_slot_api_state_errors(device, served):
    from emmax._lib import EmmaxError

    cfg, model = served
    eng = model.engine
    frames, rows = _requests(cfg, [2])
    model.generate_ids(rows, frames_u8=torch.from_numpy(frames).to(device), max_new_tokens=4)   # leaves slot mode
    with pytest.raises(EmmaxError):
        eng.slots_step(1)
    with pytest.raises(EmmaxError):
        eng.slots_open(9)
    eng.slots_open(2)
    with pytest.raises(EmmaxError):
        eng.slot_prefill(5, rows[0], None, 8)
    with pytest.raises(EmmaxError):
        eng.set_stop(list(range(17)), 1)
