"""GPU parity for the grouped-query configuration (Hq = 4, Hkv = 2: the decode attention kernel's G = 2 instance, the
fused qkv projection with fewer K/V heads, GQA flash attention in the prefill) against the CPU oracle, plus the size
limits and error behaviour of the C ABI (context overflow stops a row, over-long prompts / over-wide batches are errors).
Tolerances as in test_e2e_gpu.py: features / logits <= 3e-2 * max|ref|, ids bit-exact on margin-boosted weights."""
import numpy as np
import pytest
import torch

from conftest import ID_BUDGET_TINY, above_id_line

pytestmark = pytest.mark.gpu

FEAT_TOL = 3e-2


def _mk(cfg, seed, planted, device, **kw):
    from emmax.modeling import EmmaXForActionPrediction
    from emmax.weights import synthetic_state_dict

    sd_bf = {k: v.to(torch.bfloat16) for k, v in synthetic_state_dict(cfg, seed=seed, planted=planted).items()}
    model = EmmaXForActionPrediction(cfg, dict(sd_bf)).to(device, **kw)
    return model, {k: v.float() for k, v in sd_bf.items()}


def _inputs(B, P, seed):
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, size=(B, 224, 224, 3), dtype=np.uint8)
    rows = [[1] + [int(x) for x in rng.integers(3, 31744, size=(P if isinstance(P, int) else P[b]) - 1)] for b in range(B)]
    return frames, rows


def rel(got, ref):
    ref = ref.float().cpu()
    return ((got.float().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-9)).item()


@pytest.fixture(scope="module")
def gqa_random(device):
    from emmax.config import EmmaXConfig

    cfg = EmmaXConfig.tiny(gqa=True)
    assert cfg.llm.num_heads == 4 and cfg.llm.num_kv_heads == 2
    model, sd_ref = _mk(cfg, 23, False, device, max_batch=4, max_prompt=40)
    return cfg, model, sd_ref


def test_gqa_prefill_logits_and_teacher_forced_decode(device, gqa_random):
    from oracle import emmax_oracle as orc

    cfg, model, sd_ref = gqa_random
    frames, rows = _inputs(2, [11, 6], seed=41)
    fr = torch.from_numpy(frames).to(device)
    out = model.forward(input_ids=rows, frames_u8=fr, use_cache=True)
    for b in range(2):
        ref, _, _ = orc.vla_prefill_logits(torch.tensor([rows[b]]), orc.preprocess_frames(frames[b:b + 1], cfg), sd_ref, cfg)
        assert rel(out.logits[b], ref[0]) < FEAT_TOL
    # cached decode (split-KV attention with 2 query heads per KV head), teacher-forced with the oracle's tokens
    T = 16
    ids_ref, trace = orc.greedy_generate(torch.tensor(rows[:1]), orc.preprocess_frames(frames[:1], cfg), sd_ref, cfg, T,
                                         eos_token_id=None, return_trace=True)
    gen = ids_ref[0, len(rows[0]):].tolist()
    eng = model.engine
    model._prefill(rows[:1], None, fr[:1], max_new=T + 1)
    worst, checked, agree = 0.0, 0, 0
    for t in range(T):
        got = eng.last_logits()[0].float().cpu()
        err = (got - trace[t]).abs().max().item()
        worst = max(worst, err / trace[t].abs().max().item())
        if above_id_line(trace[t], ID_BUDGET_TINY):   # the a-priori id line (conftest.py)
            checked += 1
            agree += int(int(got.argmax()) == gen[t])
        eng.set_current_tokens([gen[t]])
        eng.decode_step()
    assert worst < FEAT_TOL, worst
    assert checked >= 1 and agree == checked, (checked, agree)


def test_gqa_planted_ids_and_batch_rows(device):
    """Margin-boosted GQA weights: ids equal the a-priori chain at bs = 1 and inside a ragged batch of 3 (MFMA path)."""
    from emmax.config import EmmaXConfig
    from emmax.weights import planted_chain, planted_start_token

    cfg = EmmaXConfig.tiny(gqa=True)
    model, _ = _mk(cfg, 5, True, device, max_batch=4, max_prompt=40)
    frames, rows = _inputs(3, [9, 14, 5], seed=8)
    for b, k in enumerate([3, 12, 7]):
        rows[b][-1] = planted_start_token(cfg, k)
    fr = torch.from_numpy(frames).to(device)
    ids_b, lens_b = model.generate_ids(rows, frames_u8=fr, max_new_tokens=32)
    for b in range(3):
        want = planted_chain(cfg, rows[b][-1], 32)
        assert ids_b[b, : int(lens_b[b])].cpu().tolist() == want
        ids_1, lens_1 = model.generate_ids(rows[b:b + 1], frames_u8=fr[b:b + 1], max_new_tokens=32)
        assert ids_1[0, : int(lens_1[0])].cpu().tolist() == want


def test_context_overflow_stops_the_row_and_limits_are_errors(device):
    """Engine-level calls (the modeling layer would silently grow the session: `ensure_capacity`)."""
    from emmax._lib import EmmaxError
    from emmax.config import EmmaXConfig
    from emmax.weights import planted_chain, planted_start_token

    cfg = EmmaXConfig.tiny()
    P = 8
    room = 6                                                   # cache slots left after the prompt
    model, _ = _mk(cfg, 5, True, device, max_batch=2, max_prompt=P, max_ctx=cfg.n_patches + P + room)
    eng = model.engine
    frames, rows = _inputs(1, P, seed=3)
    rows[0][-1] = planted_start_token(cfg, 20)                 # the chain would run for 29 tokens
    fr = torch.from_numpy(frames).to(device)
    patches = eng.vision_encode(fr)
    eng.prefill(rows, patches)
    ids, lens = eng.generate(room + 4, True)
    n = int(lens[0])
    assert 1 <= n <= room                                      # the row stops before its next append would overflow the cache
    assert ids[0, :n].cpu().tolist() == planted_chain(cfg, rows[0][-1], 40)[:n]
    assert (ids[0, n:] == cfg.pad_token_id).all()
    with pytest.raises(EmmaxError):                            # prompt longer than the session allows
        eng.prefill([[1] + [5] * 11], patches)
    with pytest.raises(EmmaxError):                            # batch wider than the session
        eng.vision_encode(torch.zeros(3, 224, 224, 3, dtype=torch.uint8, device=device))
    with pytest.raises(EmmaxError):                            # token budget beyond the output buffer
        eng.prefill(rows, patches)
        eng.generate(10 ** 6, True)
    with pytest.raises(EmmaxError):                            # empty prompt
        eng.prefill([[]], patches)
    # the session is still usable after the errors
    eng.prefill(rows, patches)
    ids2, lens2 = eng.generate(room + 4, True)
    assert ids2[0, : int(lens2[0])].cpu().tolist() == ids[0, :n].cpu().tolist()
    # and the modeling layer grows the session instead of failing
    ids3, lens3 = model.generate_ids(rows, frames_u8=fr, max_new_tokens=20)
    assert int(lens3[0]) == 20 and ids3[0].cpu().tolist() == planted_chain(cfg, rows[0][-1], 40)[:20]
