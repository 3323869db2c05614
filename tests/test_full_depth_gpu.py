"""FULL DEPTH on RANDOM weights against the fp32 oracle: the complete Emma-X-7B shape -- both ViT towers at
their real dimensions, the projector, all 32 LLaMA-2-7B layers, vocabulary 32064 -- with plain random weights (no planted
margin, no damped residual branches), one 224x224 frame, one 512-token prompt:

  * the prefill's last-position logits (768 packed rows through 32 layers of GEMM / flash attention kernels), then
  * 12 teacher-forced cached decode steps at contexts 768 .. 779 (the batch-1 K-split GEMV path, split-KV paged attention), and
  * the same request replicated to batch 8 (the MFMA small-batch path at full depth), 3 steps,

each step's logits compared with the oracle's, which runs the same bf16-rounded weights in fp32 on the host (about one minute
on the GPU box's cores).  This is where bf16 rounding error accumulated over 32 real layers is measured on the HIP path: the
bound below is what decides whether token ids can be exact on a real checkpoint -- an id can only be trusted where the oracle's
top-2 margin exceeds the logit error, so the argmax is asserted exactly there (margin > 2 x measured error) and the measured
numbers are printed (pytest -s).  Follows /root/reference/prismatic/models/vlms/prismatic.py:627-664 (generate_actions ->
generate), /root/reference/prismatic/extern/hf/modeling_prismatic.py:325-415 (cached and multimodal branches)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T_B1 = 12
T_B8 = 3
# measured on MI355X (round 3, first run): worst |err| / max|ref| 2.5e-2 .. 3.4e-2 per step at B = 1 -- sqrt(32 / 2) x the 5e-3 of the
# 2-layer tests, i.e. bf16 rounding noise adding up in quadrature over the layers (random weights: no structure damps it).  The
# bound leaves < 2x for other seeds.  What it means for token ids: see DESIGN.md section 2.
TOL = 6e-2


@pytest.fixture(scope="module")
def full(device):
    from emmax.config import EmmaXConfig
    from emmax.modeling import EmmaXForActionPrediction
    from emmax.weights import synthetic_state_dict

    cfg = EmmaXConfig.emma_x_7b()
    # CPU generator (the oracle and the device must see the same values), rounded to bf16 once; fp32 copy for the oracle
    sd_ref = synthetic_state_dict(cfg, seed=33)
    for k in sd_ref:
        sd_ref[k] = sd_ref[k].to(torch.bfloat16).float()
    sd_bf = {k: v.to(torch.bfloat16) for k, v in sd_ref.items()}
    model = EmmaXForActionPrediction(cfg, dict(sd_bf)).to(device, max_batch=8, max_prompt=512, max_ctx=256 + 512 + 32)
    return cfg, model, sd_ref, sd_bf


@pytest.fixture(scope="module")
def oracle_trace(full):
    """Greedy ids and last-position logits of T_B1 steps (step 0 = the prefill) from the fp32 oracle, one bs = 1 run."""
    from oracle import emmax_oracle as orc

    cfg, _, sd_ref, _ = full
    rng = np.random.default_rng(77)
    frames = rng.integers(0, 256, size=(1, 224, 224, 3), dtype=np.uint8)
    row = [1] + [int(x) for x in rng.integers(3, 31744, size=511)]
    torch.set_num_threads(min(64, torch.get_num_threads()))
    with torch.inference_mode():
        proj = orc.projector(orc.vision_backbone(orc.preprocess_frames(frames, cfg), sd_ref, cfg), sd_ref)
        emb = orc.splice(torch.tensor([row]), proj, sd_ref)
        logits, cache = orc.llama_forward(emb, sd_ref, cfg.llm, None, last_only=True)
        gen, trace = [], []
        for _ in range(T_B1):
            last = logits[0, -1].float()
            trace.append(last.clone())
            gen.append(int(last.argmax()))
            logits, cache = orc.llama_forward(orc.embed_tokens(torch.tensor([[gen[-1]]]), sd_ref), sd_ref, cfg.llm, cache)
    return frames, row, gen, trace


@pytest.fixture(scope="module")
def bf16_yardstick(full, oracle_trace):
    """How far a bf16 EXECUTION of the same network sits from the fp32 oracle: the oracle's bf16-emulating mode (every torch op
    rounds to bf16 = the reference's own accelerator execution, `torch_dtype=torch.bfloat16`), teacher-forced with the fp32
    oracle's ids, same frame and prompt.  Per step: max |logit error| / max |fp32 logit|."""
    from oracle import emmax_oracle as orc

    cfg, _, _, sd_bf = full
    frames, row, gen, trace = oracle_trace
    bf = torch.bfloat16
    errs = []
    with torch.inference_mode():
        proj = orc.projector(orc.vision_backbone(orc.preprocess_frames(frames, cfg).to(bf), sd_bf, cfg, dtype=bf), sd_bf, dtype=bf)
        emb = orc.splice(torch.tensor([row]), proj, sd_bf, dtype=bf)
        logits, cache = orc.llama_forward(emb, sd_bf, cfg.llm, None, dtype=bf, last_only=True)
        for t in range(T_B1):
            errs.append(((logits[0, -1].float() - trace[t]).abs().max() / trace[t].abs().max()).item())
            if t + 1 < T_B1:
                logits, cache = orc.llama_forward(orc.embed_tokens(torch.tensor([[gen[t]]]), sd_bf, dtype=bf), sd_bf, cfg.llm, cache, dtype=bf)
    return errs


def _run(model, frames, row, gen, trace, B, T, device):
    eng = model.engine
    fr = torch.from_numpy(np.repeat(frames, B, axis=0)).to(device)
    model._prefill([list(row) for _ in range(B)], None, fr, max_new=T + 1)
    worst, per_step, checked, agree, min_margin_checked = 0.0, [], 0, 0, float("inf")
    for t in range(T):
        got = eng.last_logits().float().cpu()
        ref = trace[t]
        scale = ref.abs().max().item()
        top2 = torch.topk(ref, 2).values
        margin = (top2[0] - top2[1]).item()
        step_worst = 0.0
        for b in range(B):
            err = (got[b] - ref).abs().max().item()
            step_worst = max(step_worst, err / scale)
            if margin > 2 * err:
                checked += 1
                agree += int(int(got[b].argmax()) == gen[t])
                min_margin_checked = min(min_margin_checked, margin / scale)
        per_step.append(step_worst)
        worst = max(worst, step_worst)
        eng.set_current_tokens([gen[t]] * B)
        eng.decode_step()
    return worst, per_step, checked, agree, min_margin_checked


def test_full_depth_random_weights_batch1(device, full, oracle_trace, bf16_yardstick):
    _, model, _, _ = full
    frames, row, gen, trace = oracle_trace
    worst, per_step, checked, agree, mm = _run(model, frames, row, gen, trace, 1, T_B1, device)
    print("\nfull depth B=1: worst |err|/max|ref| per step:", " ".join(f"{v:.2e}" for v in per_step),
          f"| argmax checked {checked}/{T_B1} agreed {agree} | smallest checked margin {mm:.2e} of max|logit|")
    print("bf16-emulating oracle against its fp32 mode, same steps:     ", " ".join(f"{v:.2e}" for v in bf16_yardstick))
    assert all(np.isfinite(per_step))
    assert worst < TOL, (worst, per_step)
    assert agree == checked
    assert checked >= 1, "no step had a top-2 margin above twice the error: the comparison says nothing about the ids"
    # the HIP path (bf16 storage, fp32 inside every kernel) must be no further from fp32 than a bf16 execution of the reference
    # itself: per step within 1.25x of the emulation's error (it is normally well inside: fewer rounding points)
    for t, (e, y) in enumerate(zip(per_step, bf16_yardstick)):
        assert e <= 1.25 * y + 2e-3, (t, e, y)


def test_full_depth_random_weights_batch8_mfma_path(device, full, oracle_trace):
    _, model, _, _ = full
    frames, row, gen, trace = oracle_trace
    worst, per_step, checked, agree, mm = _run(model, frames, row, gen, trace, 8, T_B8, device)
    print("\nfull depth B=8: worst |err|/max|ref| per step:", " ".join(f"{v:.2e}" for v in per_step),
          f"| argmax checked {checked}/{8 * T_B8} agreed {agree}")
    assert all(np.isfinite(per_step))
    assert worst < TOL, (worst, per_step)
    assert agree == checked
