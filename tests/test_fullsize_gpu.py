"""Full-size (Emma-X-7B dims, BASELINE.json configs 2/3 shapes) checks on a real MI355X through size-independent
properties -- the CPU oracle cannot run 7B in test time, so these use invariants instead of a reference run:

  * known answer: with the margin-boosted (planted) 7B weights greedy decoding must emit the planted successor chain
    (29871 -> 8 action tokens -> EOS) -- the same construction the tiny model is checked with against the oracle and
    against the reference wrapper's golden;
  * every row of a ragged batch == its bs=1 run (ids exact), across the dot2 (B<=2) and MFMA (B>=3) decode paths;
  * hipGraph replay == eager stepping;  run-to-run determinism;
  * the prefill's own last-position logits (MFMA GEMM path, fp32 out) == the decode-path lm-head on the same hidden
    state (GEMV path) within bf16 tolerance, and both give the same argmax;
  * action vector == de-tokenised planted ids within 1e-3."""

import numpy as np
import pytest
import torch

from conftest import ID_BUDGET_SHALLOW, above_id_line

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(device):
    from emmax.config import EmmaXConfig
    from emmax.modeling import EmmaXForActionPrediction

    cfg = EmmaXConfig.emma_x_7b()
    model = EmmaXForActionPrediction.from_synthetic(cfg, seed=0, device=device, planted=True, max_batch=4, max_prompt=512,
                                                    max_ctx=256 + 512 + 64 + 1)
    return cfg, model


def _rows(cfg, lens, steps_to_prefix, seed=3):
    from emmax.weights import planted_start_token

    rng = np.random.default_rng(seed)
    rows = []
    for n, k in zip(lens, steps_to_prefix):
        r = [1] + [int(x) for x in rng.integers(3, 31744, size=n - 1)]
        r[-1] = planted_start_token(cfg, k)
        rows.append(r)
    frames = torch.from_numpy(rng.integers(0, 256, size=(len(lens), 224, 224, 3), dtype=np.uint8))
    return frames, rows


def test_fullsize_planted_known_answer_and_actions(device, big):
    from emmax.actions import token_ids_to_actions, unnormalize
    from emmax.weights import planted_chain

    cfg, model = big
    frames, rows = _rows(cfg, [512], [20])
    acts, ids, lens = model.generate_actions_batch(frames.to(device), rows, max_new_tokens=48)
    got = ids[0, : int(lens[0])].cpu().tolist()
    want = planted_chain(cfg, rows[0][-1], 48)
    assert got == want and got[-1] == cfg.eos_token_id and len(got) == 20 + 8 + 1
    a_ids = [t for t in want if 31744 <= t < 32000][:7]
    ref = unnormalize(token_ids_to_actions(np.array(a_ids), 32000, model.bin_centers), cfg.norm_stats["bridge_orig"]["action"])
    assert np.abs(acts[0] - ref).max() <= 1e-3


def test_fullsize_ragged_batch_equals_bs1_and_is_deterministic(device, big):
    cfg, model = big
    frames, rows = _rows(cfg, [512, 37, 300, 128], [5, 12, 30, 2], seed=11)
    fr = frames.to(device)
    _, ids_b, lens_b = model.generate_actions_batch(fr, rows, max_new_tokens=40)       # B=4: MFMA decode path
    _, ids_b2, lens_b2 = model.generate_actions_batch(fr, rows, max_new_tokens=40)
    assert torch.equal(ids_b, ids_b2) and torch.equal(lens_b, lens_b2)                 # run-to-run determinism
    ids_b, lens_b = ids_b.cpu(), lens_b.cpu().tolist()
    for b in range(4):
        _, ids_1, lens_1 = model.generate_actions_batch(fr[b:b + 1].contiguous(), [rows[b]], max_new_tokens=40)   # dot2 path
        assert lens_b[b] == int(lens_1[0])
        assert ids_b[b, : lens_b[b]].tolist() == ids_1[0, : lens_b[b]].cpu().tolist()


def test_fullsize_graph_equals_eager(device, big, tune):
    tune(graph=1)
    cfg, model = big
    frames, rows = _rows(cfg, [200, 64], [25, 40], seed=5)
    fr = frames.to(device)
    T = 32
    _, ids_g, lens_g = model.generate_actions_batch(fr, rows, max_new_tokens=T)
    assert model.engine.graph_active()
    model._prefill(rows, None, fr, max_new=T)
    for _ in range(T - 1):
        model.engine.decode_step()
    ids_e, lens_e = model.engine.generate(T, True)
    assert torch.equal(ids_g, ids_e) and torch.equal(lens_g, lens_e)


def test_fullsize_graph_replay_on_the_mfma_streamk_path(device, big, tune):
    """B = 4 at 7B shapes: the small-batch MFMA projections split their work stream-K (tasks straddle two blocks and meet
    through tagged granules that the finishing block clears).  A captured step replays the same launches 40 times: if a
    granule survived a launch, or a replay read one too early, the ids would leave the eager run's."""
    cfg, model = big
    frames, rows = _rows(cfg, [300, 64, 128, 33], [20, 33, 9, 28], seed=17)
    fr = frames.to(device)
    tune(graph=1)
    _, ids_g, lens_g = model.generate_actions_batch(fr, rows, max_new_tokens=40)
    assert model.engine.graph_active()
    tune(graph=0)
    _, ids_e, lens_e = model.generate_actions_batch(fr, rows, max_new_tokens=40)
    assert not model.engine.graph_active()
    assert torch.equal(ids_g, ids_e) and torch.equal(lens_g, lens_e)
    # the whole-task split (EMMAX_STREAMK=0) sums the same products in a different fp32 order: not bit-identical, so it is
    # compared numerically -- both runs teacher-forced along the eager ids, last-position logits every step.  On this planted 7B
    # model ANY change of summation order moves the logits by 0.5-3 % of max|logit| (a flipped bf16 rounding of an activation
    # is amplified through 32 layers: stream-K vs whole tasks, stream-K vs the bs=1 dot2 path and whole tasks vs bs=1 all
    # measure the same, tools/dbg_streamk.py), so: within 5e-2, and argmax equal wherever the top-2 margin exceeds twice the
    # difference
    def forced_logits():
        model._prefill(rows, None, fr, max_new=41)
        out = []
        for t in range(24):
            out.append(model.engine.last_logits().float().cpu())
            model.engine.set_current_tokens([int(ids_e[b, min(t, int(lens_e[b]) - 1)]) for b in range(4)])
            model.engine.decode_step()
        return out

    with_sk = forced_logits()
    tune(streamk=0)
    model.engine.new_session(model.engine.max_batch, model.engine.max_prompt, model.engine.max_ctx)
    without = forced_logits()
    for a, w in zip(with_sk, without):
        diff = (a - w).abs().amax(dim=1)
        assert (diff / a.abs().amax(dim=1)).max().item() < 5e-2
        top2 = torch.topk(a, 2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 2 * diff
        assert torch.equal(a.argmax(dim=1)[clear], w.argmax(dim=1)[clear])
    tune(streamk=1)
    model.engine.new_session(model.engine.max_batch, model.engine.max_prompt, model.engine.max_ctx)


def test_fullsize_prefill_logits_consistent_with_decode_head(device, big):
    cfg, model = big
    frames, rows = _rows(cfg, [96], [3], seed=9)
    out = model.forward(input_ids=rows, frames_u8=frames.to(device))
    full = out.logits[0]                                  # [S, vocab] from the MFMA GEMM (fp32 out)
    assert full.shape == (256 + 96, cfg.llm.vocab_size)
    last = model.engine.last_logits()[0]                  # lm-head GEMV on the gathered last row
    err = (full[-1] - last).abs().max().item() / last.abs().max().item()
    assert err < 1e-2
    assert int(full[-1].argmax()) == int(last.argmax())
    assert torch.isfinite(full).all()


def test_fullsize_vision_towers_match_oracle(device):
    """The two ViT towers at their REAL sizes (DINOv2-L/14 reg4: D 1024, 24 blocks, 16 heads of 64; SigLIP so400m: D 1152,
    27 blocks, 16 heads of 72, MLP 4304 -> padded 4352) + the fused projector, against the fp32 CPU oracle on the same
    bf16-rounded random weights.  The language model is the tiny one (the 7B LLM is covered above and would not fit the
    CPU oracle).  Tolerance 3e-2 * max|ref| after 23 + 26 fused bf16 blocks (as in test_e2e_gpu.py)."""
    from emmax.config import EmmaXConfig
    from emmax.modeling import EmmaXForActionPrediction
    from emmax.weights import synthetic_state_dict
    from oracle import emmax_oracle as orc

    big, tiny = EmmaXConfig.emma_x_7b(), EmmaXConfig.tiny()
    cfg = EmmaXConfig(big.towers, tiny.llm, norm_stats=tiny.norm_stats)
    sd = synthetic_state_dict(cfg, seed=3)                       # CPU generator: the oracle and the device see the same values
    sd_bf = {k: v.to(torch.bfloat16) for k, v in sd.items()}
    model = EmmaXForActionPrediction(cfg, dict(sd_bf)).to(device, max_batch=2, max_prompt=16)
    sd_ref = {k: v.float() for k, v in sd_bf.items() if k.startswith(("vision_backbone.", "projector."))}
    rng = np.random.default_rng(12)
    frames = rng.integers(0, 256, size=(2, 224, 224, 3), dtype=np.uint8)
    got_proj = model.engine.vision_encode(torch.from_numpy(frames).to(device))
    got_feats = model.engine.vision_features(2)
    pix = orc.preprocess_frames(frames, cfg)
    ref_feats = orc.vision_backbone(pix, sd_ref, cfg)
    ref_proj = orc.projector(ref_feats, sd_ref)
    assert got_feats.shape[-1] >= 1024 + 1152 and got_proj.shape == (2, 256, cfg.llm.hidden_size)

    def rel(a, b):
        b = b.float().cpu()
        return ((a.float().cpu() - b).abs().max() / b.abs().max()).item()

    assert rel(got_feats[..., : ref_feats.shape[-1]], ref_feats) < 3e-2
    assert rel(got_proj, ref_proj) < 3e-2
    # batch invariance: frame 1 alone gives the same features as inside the batch of 2, up to the fp32 summation order of the
    # tile / split-K plan (a lone frame takes the split-K path for fc2); far inside the bf16 tolerance
    alone = model.engine.vision_encode(torch.from_numpy(frames[1:2]).to(device))
    assert rel(alone[0], got_proj[1]) < 1e-2


def test_fullsize_vision_towers_large_batch_plans(device):
    """BASELINE configs[3] (ViT-only, batch 256): at B >= 16 `launch_gemm` takes the big-tile / row-split plans instead of the
    small-tile and split-K plans of B <= 2.  B = 16 at the real tower dimensions against the fp32 oracle (3e-2 * max|ref| as
    above), then B = 256 (the 16 frames tiled 16x) against the oracle and the B = 16 result of the same frames.  At B = 256 both
    towers also take the RESIDENT attention kernel (4096 (frame, head) items: whole K / V of an item in LDS, producer waves,
    lazy rescale) instead of the ring kernel of the small batches -- this is its end-to-end check against the fp32 oracle
    (`test_attention_resident_form_many_short_sequences` is the op-level one)."""
    from emmax.config import EmmaXConfig
    from emmax.modeling import EmmaXForActionPrediction
    from emmax.weights import synthetic_state_dict
    from oracle import emmax_oracle as orc

    big, tiny = EmmaXConfig.emma_x_7b(), EmmaXConfig.tiny()
    cfg = EmmaXConfig(big.towers, tiny.llm, norm_stats=tiny.norm_stats)
    sd_bf = {k: v.to(torch.bfloat16) for k, v in synthetic_state_dict(cfg, seed=4).items()}
    model = EmmaXForActionPrediction(cfg, dict(sd_bf)).to(device, max_batch=256, max_prompt=8)
    sd_ref = {k: v.float() for k, v in sd_bf.items() if k.startswith(("vision_backbone.", "projector."))}
    rng = np.random.default_rng(31)
    frames = rng.integers(0, 256, size=(16, 224, 224, 3), dtype=np.uint8)
    got_proj = model.engine.vision_encode(torch.from_numpy(frames).to(device)).float().cpu()
    got_feats = model.engine.vision_features(16).float().cpu()
    with torch.inference_mode():
        ref_feats = orc.vision_backbone(orc.preprocess_frames(frames, cfg), sd_ref, cfg)
        ref_proj = orc.projector(ref_feats, sd_ref)

    def rel(a, b):
        return ((a - b).abs().max() / b.abs().max()).item()

    def fro(a, b):
        return ((a - b).norm() / b.norm()).item()

    # against the fp32 oracle.  Frobenius norm: the residual stream is stored in bf16 (like the reference's own bf16 run) and
    # rounded twice per block -- a random walk of 2 x 23 roundings of relative size 2^-9 / sqrt(3) gives ~1.5e-2 (measured
    # 1.2e-2): bound 2e-2.  The MAX over the 8.9 M feature values of 16 frames reaches further into the rounding tail than the
    # 2-frame test above (measured 3.4e-2): 5e-2.
    gf = got_feats[..., : ref_feats.shape[-1]]
    assert fro(gf, ref_feats) < 2e-2 and fro(got_proj, ref_proj) < 2e-2, (fro(gf, ref_feats), fro(got_proj, ref_proj))
    assert rel(gf, ref_feats) < 5e-2 and rel(got_proj, ref_proj) < 5e-2, (rel(gf, ref_feats), rel(got_proj, ref_proj))
    # against the launch plans the 2-frame test pins (small tiles / split-K): the same frames in batches of 2.  A different tile
    # plan changes the fp32 summation order, which flips bf16 roundings of intermediate activations; over 49 blocks two plans
    # drift as far from each other as each is from the oracle (measured 1.3e-2 Frobenius, 1.4e-2 of max|ref|): same bounds
    for i in range(0, 16, 2):
        pair = model.engine.vision_encode(torch.from_numpy(frames[i:i + 2]).to(device)).float().cpu()
        assert rel(pair, got_proj[i:i + 2]) < 5e-2 and fro(pair, got_proj[i:i + 2]) < 2e-2, (i, rel(pair, got_proj[i:i + 2]), fro(pair, got_proj[i:i + 2]))
        assert rel(pair, ref_proj[i:i + 2]) < 5e-2 and fro(pair, ref_proj[i:i + 2]) < 2e-2, i
    big_b = torch.from_numpy(np.tile(frames, (16, 1, 1, 1))).to(device)
    got256 = model.engine.vision_encode(big_b).float().cpu()
    assert got256.shape[0] == 256
    ref256 = ref_proj.repeat(16, 1, 1)
    assert fro(got256, ref256) < 2e-2 and rel(got256, ref256) < 5e-2, (fro(got256, ref256), rel(got256, ref256))
    for r in range(16):
        assert rel(got256[16 * r:16 * r + 16], got_proj) < 5e-2 and fro(got256[16 * r:16 * r + 16], got_proj) < 2e-2, r


@pytest.fixture(scope="module")
def llm2(device):
    """LLaMA-2-7B layer dimensions, 2 layers, tiny towers, random weights (CPU generator: oracle and device see the same)."""
    from emmax.config import EmmaXConfig, LlmConfig
    from emmax.modeling import EmmaXForActionPrediction
    from emmax.weights import synthetic_state_dict

    tiny = EmmaXConfig.tiny()
    llm = LlmConfig(hidden_size=4096, intermediate_size=11008, num_layers=2, num_heads=32, num_kv_heads=32, head_dim=128,
                    vocab_size=32064, max_position=2048)
    cfg = EmmaXConfig(tiny.towers, llm, norm_stats=tiny.norm_stats)
    sd_bf = {k: v.to(torch.bfloat16) for k, v in synthetic_state_dict(cfg, seed=9).items()}
    model = EmmaXForActionPrediction(cfg, dict(sd_bf)).to(device, max_batch=4, max_prompt=32)
    return cfg, model, {k: v.float() for k, v in sd_bf.items()}


def _teacher_forced(model, cfg, sd_ref, frames, rows, T, device):
    """Per-row worst relative logit error and (checked, agreed) argmax counts over T teacher-forced decode steps of a batch."""
    from oracle import emmax_oracle as orc

    B = len(rows)
    gens, traces = [], []
    for b in range(B):
        ids_ref, trace = orc.greedy_generate(torch.tensor(rows[b:b + 1]), orc.preprocess_frames(frames[b:b + 1], cfg), sd_ref, cfg, T,
                                             eos_token_id=None, return_trace=True)
        gens.append(ids_ref[0, len(rows[b]):].tolist())
        traces.append(trace)
    eng = model.engine
    model._prefill(rows, None, torch.from_numpy(frames).to(device), max_new=T + 1)
    worst, checked, agree = 0.0, 0, 0
    for t in range(T):
        got = eng.last_logits().float().cpu()
        for b in range(B):
            ref = traces[b][t]
            err = (got[b] - ref).abs().max().item()
            worst = max(worst, err / ref.abs().max().item())
            if above_id_line(ref, ID_BUDGET_SHALLOW):   # the a-priori id line (conftest.py), not twice the measured error
                checked += 1
                agree += int(int(got[b].argmax()) == gens[b][t])
        eng.set_current_tokens([gens[b][t] for b in range(B)])
        eng.decode_step()
    return worst, checked, agree


def test_fullsize_llm_dims_two_layers_match_oracle(device, llm2):
    """LLaMA-2-7B layer dimensions (hidden 4096, 32 heads of 128, intermediate 11008, vocab 32064) with 2 layers and the tiny
    towers, random weights, against the fp32 CPU oracle: every prefill logit row (the GEMM launch plans / split-K of the real
    shapes) and 8 teacher-forced decode steps (the GEMV / paged-attention kernels at their real K and N).  Tolerances as in
    test_e2e_gpu.py: 3e-2 * max|ref|; argmax equal wherever the oracle's top-2 margin clears the a-priori id line (conftest.py)."""
    from oracle import emmax_oracle as orc

    cfg, model, sd_ref = llm2
    rng = np.random.default_rng(4)
    frames = rng.integers(0, 256, size=(1, 224, 224, 3), dtype=np.uint8)
    rows = [[1] + [int(x) for x in rng.integers(3, 31744, size=13)]]
    out = model.forward(input_ids=rows, frames_u8=torch.from_numpy(frames).to(device), use_cache=True)
    ref, _, _ = orc.vla_prefill_logits(torch.tensor(rows), orc.preprocess_frames(frames, cfg), sd_ref, cfg)
    got = out.logits[0].float().cpu()
    assert got.shape == ref[0].shape == (256 + 14, 32064)
    assert ((got - ref[0]).abs().max() / ref[0].abs().max()).item() < 3e-2
    worst, checked, agree = _teacher_forced(model, cfg, sd_ref, frames, rows, 8, device)
    assert worst < 3e-2, worst
    assert agree == checked


def test_fullsize_llm_dims_batch3_mfma_path_matches_oracle(device, llm2):
    """The same at batch 3 (ragged prompts): the MFMA small-batch projections over the fragment-major weight copy and the
    batched split-KV attention at the real dimensions, every row against its own oracle run."""
    cfg, model, sd_ref = llm2
    rng = np.random.default_rng(6)
    frames = rng.integers(0, 256, size=(3, 224, 224, 3), dtype=np.uint8)
    rows = [[1] + [int(x) for x in rng.integers(3, 31744, size=n)] for n in (9, 17, 5)]
    worst, checked, agree = _teacher_forced(model, cfg, sd_ref, frames, rows, 5, device)
    assert worst < 3e-2, worst
    assert checked >= 2 and agree == checked


def _dequant_e4m3_rows(w: torch.Tensor) -> torch.Tensor:
    """What the fp8 decode path computes with: per-row scale amax/448, weights rounded to OCP e4m3 (RNE), back in fp32."""
    w = w.float()
    scale = (w.abs().amax(dim=1, keepdim=True) / 448.0).clamp_min(1e-30)
    return (w / scale).to(torch.float8_e4m3fn).float() * scale


def test_fullsize_llm_dims_fp8_decode_matches_dequantised_oracle(device):
    """BASELINE config 5 at the LLaMA-2-7B layer dimensions (2 layers): the fp8-e4m3 decode path against the fp32 oracle run
    on the DE-QUANTISED weights (prefill keeps bf16 weights, as the device does; the lm-head and every decode projection see
    e4m3 x per-row scale).  Not "close to bf16" but parity with the arithmetic the path claims to perform: 3e-2 * max|ref|,
    argmax equal wherever the oracle's margin exceeds 2x the measured error; B = 1 and 2 (dot-product GEMV over the e4m3 rows,
    MFMA kernel for the down projection) and a ragged batch of 3 (MFMA kernel throughout)."""
    import copy

    from emmax.config import EmmaXConfig, LlmConfig
    from emmax.modeling import EmmaXForActionPrediction
    from emmax.weights import synthetic_state_dict
    from oracle import emmax_oracle as orc

    tiny = EmmaXConfig.tiny()
    llm = LlmConfig(hidden_size=4096, intermediate_size=11008, num_layers=2, num_heads=32, num_kv_heads=32, head_dim=128,
                    vocab_size=32064, max_position=2048)
    cfg = EmmaXConfig(tiny.towers, llm, norm_stats=tiny.norm_stats)
    cfg8 = copy.deepcopy(cfg)
    cfg8.decode_weight_dtype = "fp8"
    sd_bf = {k: v.to(torch.bfloat16) for k, v in synthetic_state_dict(cfg, seed=9).items()}
    model = EmmaXForActionPrediction(cfg8, dict(sd_bf)).to(device, max_batch=4, max_prompt=32)
    sd_ref = {k: v.float() for k, v in sd_bf.items()}
    proj = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")
    sd_q = {k: (_dequant_e4m3_rows(v) if (any(p in k for p in proj) or k.endswith("lm_head.weight")) else v) for k, v in sd_ref.items()}
    sd_prefill = dict(sd_ref)
    sd_prefill["language_model.lm_head.weight"] = sd_q["language_model.lm_head.weight"]

    rng = np.random.default_rng(8)
    frames = rng.integers(0, 256, size=(3, 224, 224, 3), dtype=np.uint8)
    rows = [[1] + [int(x) for x in rng.integers(3, 31744, size=n)] for n in (11, 6, 15)]
    T = 5
    for sel in ([0], [2, 1], [0, 1, 2]):
        fr, rr = frames[sel], [rows[i] for i in sel]
        gens, traces = [], []
        for b in range(len(sel)):
            logits, cache, _ = orc.vla_prefill_logits(torch.tensor([rr[b]]), orc.preprocess_frames(fr[b:b + 1], cfg), sd_prefill, cfg)
            gen, trace = [], []
            for _ in range(T):
                last = logits[0, -1].float()
                trace.append(last.clone())
                gen.append(int(last.argmax()))
                logits, cache = orc.llama_forward(orc.embed_tokens(torch.tensor([[gen[-1]]]), sd_q), sd_q, cfg.llm, cache)
            gens.append(gen)
            traces.append(trace)
        eng = model.engine
        model._prefill(rr, None, torch.from_numpy(fr).to(device), max_new=T + 1)
        worst, checked, agree = 0.0, 0, 0
        for t in range(T):
            got = eng.last_logits().float().cpu()
            for b in range(len(sel)):
                ref = traces[b][t]
                err = (got[b] - ref).abs().max().item()
                worst = max(worst, err / ref.abs().max().item())
                if above_id_line(ref, 2 * ID_BUDGET_SHALLOW):   # fp8 weights: twice the bf16 budget, fixed before the run
                    checked += 1
                    agree += int(int(got[b].argmax()) == gens[b][t])
            eng.set_current_tokens([gens[b][t] for b in range(len(sel))])
            eng.decode_step()
        assert worst < 3e-2, (sel, worst)
        assert agree == checked, (sel, agree, checked)
