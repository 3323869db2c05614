"""How far is a bf16 execution from the fp32 CPU reference path?  (SURVEY.md section 7 hard part (ii); VERDICT r01 missing #5)

The oracle runs the same op graph in two modes: `dtype=float32` (the reference's CPU path, the parity target) and
`dtype=bfloat16` (every torch op rounds its output to bf16: what the reference itself does on an accelerator with
torch_dtype=bfloat16).  On RANDOM weights (thin top-1 margins, the adversarial case) over 64 teacher-forced greedy steps:
  * CPU test: the bf16-emulating oracle against the fp32 oracle -- logit error and the number of steps whose argmax flips;
    every flip must sit on a step whose fp32 top-2 margin is below 2x that step's logit error (margin-aware exactness).
  * GPU test: the HIP path against the fp32 oracle on the same run -- it keeps fp32 inside the fused kernels (softmax, P.V,
    SwiGLU, RoPE, residual adds before the bf16 store), so its logit error must not exceed the bf16-emulating oracle's, and
    it may flip only where the margin rule allows.  The counts are printed (pytest -s) and quoted in DESIGN.md."""

import numpy as np
import pytest
import torch

T = 64


def _setup(seed=11):
    from emmax.config import EmmaXConfig
    from emmax.weights import synthetic_state_dict

    cfg = EmmaXConfig.tiny()
    sd_bf = {k: v.to(torch.bfloat16) for k, v in synthetic_state_dict(cfg, seed=seed).items()}
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, size=(1, 224, 224, 3), dtype=np.uint8)
    row = [1] + [int(x) for x in rng.integers(3, 31744, size=20)]
    return cfg, sd_bf, frames, row


def _fp32_run(cfg, sd_ref, frames, row):
    from oracle import emmax_oracle as orc

    ids, trace = orc.greedy_generate(torch.tensor([row]), orc.preprocess_frames(frames, cfg), sd_ref, cfg, T, eos_token_id=None,
                                     return_trace=True)
    return ids[0, len(row):].tolist(), trace


def _bf16_oracle_teacher_forced(cfg, sd_bf, frames, row, gen):
    from oracle import emmax_oracle as orc

    bf = torch.bfloat16
    logits, cache, _ = orc.vla_prefill_logits(torch.tensor([row]), orc.preprocess_frames(frames, cfg), sd_bf, cfg, bf)
    out = []
    for t in range(T):
        out.append(logits[0, -1].float())
        logits, cache = orc.llama_forward(orc.embed_tokens(torch.tensor([[gen[t]]]), sd_bf, bf), sd_bf, cfg.llm, cache, bf)
    return out


def _stats(got, trace, gen):
    errs, flips, illegal = [], 0, 0
    for t in range(T):
        ref = trace[t]
        err = (got[t] - ref).abs().max().item()
        errs.append(err / ref.abs().max().item())
        if int(got[t].argmax()) != gen[t]:
            flips += 1
            top2 = torch.topk(ref, 2).values
            illegal += int((top2[0] - top2[1]).item() > 2 * err)
    return max(errs), float(np.mean(errs)), flips, illegal


def test_bf16_emulating_oracle_vs_fp32_oracle():
    cfg, sd_bf, frames, row = _setup()
    sd_ref = {k: v.float() for k, v in sd_bf.items()}
    with torch.inference_mode():
        gen, trace = _fp32_run(cfg, sd_ref, frames, row)
        worst, mean, flips, illegal = _stats(_bf16_oracle_teacher_forced(cfg, sd_bf, frames, row, gen), trace, gen)
    print(f"bf16-emulating oracle vs fp32 oracle over {T} steps: worst rel logit err {worst:.4f}, mean {mean:.4f}, argmax flips {flips}")
    assert worst < 8e-2          # a per-op-rounded bf16 run of this model sits at a few percent of max|logit|
    assert illegal == 0          # every flip is a near-tie


@pytest.mark.gpu
def test_hip_path_is_no_further_from_fp32_than_a_bf16_reference_run(device):
    from emmax.modeling import EmmaXForActionPrediction

    cfg, sd_bf, frames, row = _setup()
    sd_ref = {k: v.float() for k, v in sd_bf.items()}
    with torch.inference_mode():
        gen, trace = _fp32_run(cfg, sd_ref, frames, row)
        o_worst, o_mean, o_flips, _ = _stats(_bf16_oracle_teacher_forced(cfg, sd_bf, frames, row, gen), trace, gen)
    model = EmmaXForActionPrediction(cfg, dict(sd_bf)).to(device, max_batch=1, max_prompt=32)
    model._prefill([row], None, torch.from_numpy(frames).to(device), max_new=T + 1)
    got = []
    for t in range(T):
        got.append(model.engine.last_logits()[0].float().cpu())
        model.engine.set_current_tokens([gen[t]])
        model.engine.decode_step()
    worst, mean, flips, illegal = _stats(got, trace, gen)
    print(f"HIP path vs fp32 oracle over {T} steps: worst rel logit err {worst:.4f}, mean {mean:.4f}, argmax flips {flips} "
          f"(bf16-emulating oracle: worst {o_worst:.4f}, mean {o_mean:.4f}, flips {o_flips})")
    assert illegal == 0
    assert worst < 3e-2
    assert mean <= o_mean * 1.25 + 1e-4, (mean, o_mean)
