"""FULL DEPTH on RANDOM weights against the fp32 oracle: the complete Emma-X-7B shape -- both ViT towers at
their real dimensions, the projector, all 32 LLaMA-2-7B layers, vocabulary 32064 -- with plain random weights (no planted
margin, no damped residual branches), one 224x224 frame, one 512-token prompt:

  * the prefill's last-position logits (768 packed rows through 32 layers of GEMM / flash attention kernels), then
  * 12 teacher-forced cached decode steps at contexts 768 .. 779 (the batch-1 K-split GEMV path, split-KV paged attention), and
  * the same request replicated to batch 8 (the MFMA small-batch path at full depth), 3 steps,

each step's logits compared with the oracle's, which runs the same bf16-rounded weights in fp32 on the host (about one minute
on the GPU box's cores).  This is where bf16 rounding error accumulated over 32 real layers is measured on the HIP path: the
bound below is what decides whether token ids can be exact on a real checkpoint.  Round 6: the argmax is asserted
wherever the oracle's top-2 margin clears an A-PRIORI id line -- twice a FIXED error budget (conftest.py: 4.5e-2 of max|logit| for the
bf16-operand path at full depth, 1e-4 for exact numerics), not twice the error this run happened to measure -- and the per-step error
against its own tolerance; the measured numbers are printed (pytest -s).  The exact-numerics session (tuning switch exact) is held to
fp32 tolerances at the end of this file: logit error <= 2e-4, ids equal on free-running 512-token generations.  Follows /root/reference/prismatic/models/vlms/prismatic.py:627-664 (generate_actions ->
generate), /root/reference/prismatic/extern/hf/modeling_prismatic.py:325-415 (cached and multimodal branches)."""

import numpy as np
import pytest
import torch

from conftest import ID_BUDGET_EXACT, ID_BUDGET_FP8_FULL_DEPTH, ID_BUDGET_FULL_DEPTH, above_id_line

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


def _run(model, frames, row, gen, trace, B, T, device, budget=ID_BUDGET_FULL_DEPTH):
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
            if above_id_line(ref, budget):      # the a-priori id line (conftest.py)
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
    assert agree == checked     # (of 12 random-weight steps few clear a 9e-2 line: the 512-step statistic below is where the line bites)
    # the HIP path (bf16 storage, fp32 inside every kernel) must be no further from fp32 than a bf16 execution of the reference
    # itself: per step within 1.25x of the emulation's error (it is normally well inside: fewer rounding points)
    for t, (e, y) in enumerate(zip(per_step, bf16_yardstick)):
        assert e <= 1.25 * y + 2e-3, (t, e, y)


@pytest.mark.parametrize("B", [8, 16, 32])
def test_full_depth_random_weights_batch8_mfma_path(device, full, oracle_trace, B):
    """B = 8: decode_km.hip; round 5: B = 16 (sixteen staged rows) and B = 32 (decode_kmp.hip: two batch tiles per weight tile) at all 32 layers."""
    _, model, _, _ = full
    frames, row, gen, trace = oracle_trace
    if B > 8:
        model.engine.new_session(B, 512, 256 + 512 + 32)
    worst, per_step, checked, agree, mm = _run(model, frames, row, gen, trace, B, T_B8, device)
    print(f"\nfull depth B={B}: worst |err|/max|ref| per step:", " ".join(f"{v:.2e}" for v in per_step),
          f"| argmax checked {checked}/{B * T_B8} agreed {agree}")
    if B > 8:
        model.engine.new_session(8, 512, 256 + 512 + 32)
    assert all(np.isfinite(per_step))
    assert worst < TOL, (worst, per_step)
    assert agree == checked


@pytest.mark.parametrize("B", [1, 8])
def test_full_depth_over_the_fp8_kv_cache(device, full, oracle_trace, tune, B):
    """VERDICT r04 next #3c: the opt-in fp8-e4m3 KV cache (tuning switch kv_fp8 at session creation) at FULL depth -- its own logit-error
    line against the fp32 oracle, next to the bf16 cache's on the same steps.  e4m3 carries 3 mantissa bits on every K and V element:
    the error is bounded at 3 x TOL here (measured ~2-3 x the bf16 cache's), and the argmax must still hold wherever the oracle's
    margin exceeds twice the measured error."""
    _, model, _, _ = full
    frames, row, gen, trace = oracle_trace
    lines = {}
    for kv8 in (0, 1):
        tune(kv_fp8=kv8)
        model.engine.new_session(8, 512, 256 + 512 + 32)     # the cache format is fixed when the session is created
        worst, per_step, checked, agree, mm = _run(model, frames, row, gen, trace, B, T_B1 if B == 1 else T_B8, device)
        lines[kv8] = per_step
        assert all(np.isfinite(per_step)) and agree == checked, (kv8, per_step, agree, checked)
    tune(kv_fp8=0)
    model.engine.new_session(8, 512, 256 + 512 + 32)
    print(f"\nfull depth B={B}, bf16 KV cache:  worst |err|/max|ref| per step:", " ".join(f"{v:.2e}" for v in lines[0]))
    print(f"full depth B={B}, e4m3 KV cache:  worst |err|/max|ref| per step:", " ".join(f"{v:.2e}" for v in lines[1]))
    assert max(lines[0]) < TOL and max(lines[1]) < 3 * TOL, lines


# ---------------------------------------------------------------------------------------------------------------------
# Round 4 (VERDICT r03 "next" #4): fp8 and RAGGED batch 8 at full depth, and how often an id COULD flip over a whole generation.
#
# The checker for these is the SAME fp32 restatement (oracle/emmax_oracle.py), executed by torch on the GPU instead of the host
# cores: 8 ragged rows x (32-layer prefill + decode steps) and a 512-step generation are ~15 minutes of host fp32 against seconds
# on the device.  `test_device_executed_oracle_equals_the_cpu_oracle` ties that execution to the CPU oracle first: same frame,
# same prompt, every one of the 12 steps of `oracle_trace` -- logits to 1e-4 of max|logit| (fp32 summation order is the only
# difference), ids equal.  Nothing here is product code: the product path stays libemmax_hip.so through ctypes.
# ---------------------------------------------------------------------------------------------------------------------
LENS8 = [572, 64, 570, 32, 128, 575, 16, 300]   # prompt tokens per row: S = 828 / 830 / 831 cross the KV page boundary at 832 in steps 4 / 2 / 1
T_R8 = 8
TOL_FP8 = 9e-2
PROJ = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


def _dequant_e4m3_rows(w):
    w = w.float()
    scale = (w.abs().amax(dim=1, keepdim=True) / 448.0).clamp_min(1e-30)
    return (w / scale).to(torch.float8_e4m3fn).float() * scale


@pytest.fixture(scope="module")
def dev_oracle(device, full):
    """fp32 weights of the oracle on the device: `ref` = the bf16-rounded values (what the bf16 model holds), `q` = the decode
    projections + lm-head as e4m3 x per-row scale de-quantised (what the fp8 model streams), `prefill_q` = bf16 weights with the
    fp8 lm-head (the fp8 model's prefill)."""
    _, _, sd_ref, _ = full
    ref = {k: v.to(device) for k, v in sd_ref.items()}
    q = {k: (_dequant_e4m3_rows(v) if (any(p in k for p in PROJ) or k.endswith("lm_head.weight")) else v) for k, v in ref.items()}
    prefill_q = dict(ref)
    prefill_q["language_model.lm_head.weight"] = q["language_model.lm_head.weight"]
    return ref, q, prefill_q


def _dev_trace(cfg, sd_prefill, sd_decode, frames, row, T, device):
    """Greedy ids + last-position logits (cpu f32) of T steps of ONE request, fp32 restatement executed on `device`."""
    from oracle import emmax_oracle as orc

    with torch.inference_mode():
        pix = orc.preprocess_frames(frames, cfg).to(device)
        proj = orc.projector(orc.vision_backbone(pix, sd_prefill, cfg), sd_prefill)
        emb = orc.splice(torch.tensor([row], device=device), proj, sd_prefill)
        logits, cache = orc.llama_forward(emb, sd_prefill, cfg.llm, None, last_only=True)
        gen, trace = [], []
        for _ in range(T):
            last = logits[0, -1].float()
            trace.append(last.cpu())
            gen.append(int(last.argmax()))
            logits, cache = orc.llama_forward(orc.embed_tokens(torch.tensor([[gen[-1]]], device=device), sd_decode), sd_decode, cfg.llm, cache)
    return gen, trace


def test_device_executed_oracle_equals_the_cpu_oracle(device, full, oracle_trace, dev_oracle):
    cfg = full[0]
    frames, row, gen, trace = oracle_trace
    ref = dev_oracle[0]
    gen_d, trace_d = _dev_trace(cfg, ref, ref, frames, row, T_B1, device)
    worst = max(((a - b).abs().max() / b.abs().max()).item() for a, b in zip(trace_d, trace))
    print(f"\nfp32 restatement on the device against on the host cores: worst |diff| / max|logit| over {T_B1} steps = {worst:.2e}")
    assert worst < 1e-4
    assert gen_d == gen


@pytest.fixture(scope="module")
def ragged8(full):
    rng = np.random.default_rng(404)
    frames = rng.integers(0, 256, size=(8, 224, 224, 3), dtype=np.uint8)
    rows = [[1] + [int(x) for x in rng.integers(3, 31744, size=n - 1)] for n in LENS8]
    return frames, rows


def _teacher_forced_rows(model, frames, rows, gens, traces, sel, T, device, budget=ID_BUDGET_FULL_DEPTH):
    eng = model.engine
    model._prefill([rows[i] for i in sel], None, torch.from_numpy(frames[sel]).to(device), max_new=T + 1)
    per_step, checked, agree = [], 0, 0
    for t in range(T):
        got = eng.last_logits().float().cpu()
        w = 0.0
        for j, i in enumerate(sel):
            ref = traces[i][t]
            err = (got[j] - ref).abs().max().item()
            w = max(w, err / ref.abs().max().item())
            if above_id_line(ref, budget):
                checked += 1
                agree += int(int(got[j].argmax()) == gens[i][t])
        per_step.append(w)
        eng.set_current_tokens([gens[i][t] for i in sel])
        eng.decode_step()
    return per_step, checked, agree


def test_full_depth_ragged_batch8_bf16(device, full, dev_oracle, ragged8):
    """configs[2]'s per-GPU shard as it really looks: eight DIFFERENT requests of different lengths at full depth (the round-3 test
    replicated one request 8x), three of them crossing the KV page boundary at 832 inside the window."""
    cfg, model, _, _ = full
    frames, rows = ragged8
    ref = dev_oracle[0]
    model.engine.new_session(8, 576, 256 + 576 + 32)
    gens, traces = zip(*[_dev_trace(cfg, ref, ref, frames[b:b + 1], rows[b], T_R8, device) for b in range(8)])
    per_step, checked, agree = _teacher_forced_rows(model, frames, rows, gens, traces, list(range(8)), T_R8, device)
    print("\nfull depth, ragged B=8, bf16: worst |err|/max|ref| per step:", " ".join(f"{v:.2e}" for v in per_step),
          f"| argmax checked {checked}/{8 * T_R8} agreed {agree}")
    assert all(np.isfinite(per_step)) and max(per_step) < TOL, per_step
    assert agree == checked


def test_full_depth_fp8_batch1_and_ragged_batch8(device, full, dev_oracle, ragged8):
    """BASELINE configs[4] at full depth (round 3 checked fp8 at 2 layers only): e4m3 decode weights with per-row scales, all 32
    layers, against the fp32 restatement run on the DE-QUANTISED weights -- B = 1 (fp8 row GEMV / K-split MFMA routing) and the
    ragged B = 8 (K-split MFMA kernels), eager launches and hipGraph replay of the step."""
    import copy

    from emmax import _lib
    from emmax.modeling import EmmaXForActionPrediction

    cfg, _, _, sd_bf = full
    frames, rows = ragged8
    _, q, prefill_q = dev_oracle
    c8 = copy.deepcopy(cfg)
    c8.decode_weight_dtype = "fp8"
    model8 = EmmaXForActionPrediction(c8, dict(sd_bf)).to(device, max_batch=8, max_prompt=576, max_ctx=256 + 576 + 32)
    gens, traces = zip(*[_dev_trace(cfg, prefill_q, q, frames[b:b + 1], rows[b], T_R8, device) for b in range(8)])
    for graph in (0, 1):
        with _lib.tuning(graph=graph):
            for sel in ([0], list(range(8))):
                per_step, checked, agree = _teacher_forced_rows(model8, frames, rows, gens, traces, sel, T_R8, device, budget=ID_BUDGET_FP8_FULL_DEPTH)
                assert model8.engine.graph_active() == bool(graph)
                print(f"\nfull depth fp8, B={len(sel)}{' ragged' if len(sel) > 1 else ''}, {'hipGraph' if graph else 'eager'}: worst |err|/max|ref| per step:",
                      " ".join(f"{v:.2e}" for v in per_step), f"| argmax checked {checked}/{len(sel) * T_R8} agreed {agree}")
                # measured (round 4): 2.6e-2 .. 6.0e-2 per step against 2.9e-2 .. 3.7e-2 for bf16 weights -- the e4m3 path keeps
                # bf16 activations but its K-split MFMA kernels round the attention output and the SwiGLU product once more
                assert all(np.isfinite(per_step)) and max(per_step) < TOL_FP8, per_step
                assert agree == checked
    del model8
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def full_exact(device, full):
    """The same weights in an EXACT-NUMERICS model (tuning switch exact at finalize + session creation): fp32 activations, two-term bf16
    operands, fp32 attention over an fp32 KV cache."""
    from emmax.modeling import EmmaXForActionPrediction

    cfg, _, _, sd_bf = full
    return EmmaXForActionPrediction(cfg, dict(sd_bf)).to(device, max_batch=1, max_prompt=512, max_ctx=256 + 512 + 512 + 8, exact=True)


TOL_EXACT = 2e-4    # exact numerics at full depth: |logit err| <= 2e-4 of max|logit| (VERDICT r05 asked <= 2e-3; measured ~1e-5)


def test_how_often_an_id_could_flip_over_a_512_token_generation(device, full, full_exact, dev_oracle, oracle_trace):
    """The statistic the parity claim rests on: over a FULL 512-step teacher-forced generation at full depth (B = 1, the bench's own step
    sequence: contexts 768 .. 1279), per step the HIP path's logit error against the fp32 restatement and the fp32 top-2 margin -- for the
    default bf16-operand path (fp32 residual stream) and for EXACT NUMERICS (round 6).  The id line is A PRIORI (conftest.py): an id must
    match wherever the margin exceeds twice the path's fixed error budget (4.5e-2 / 1e-4 of max|logit|); steps below the line are counted,
    and how many of them actually flipped.  Written to gpurun_out/r06_margin_statistic.json (committed as profiles/r06_margin_statistic.json).

    Asserted: no flip above the line, error within tolerance -- and for exact numerics the marks VERDICT r05 set: median error <= 2e-3
    (asserted at 2e-4), at most 2 flips in 512 steps."""
    import json
    import os

    from conftest import ROOT

    cfg, model, _, _ = full
    frames, row, _, _ = oracle_trace
    ref = dev_oracle[0]
    T = 512
    gen, trace = _dev_trace(cfg, ref, ref, frames, row, T, device)

    def run_mode(mdl, budget):
        eng = mdl.engine
        mdl._prefill([list(row)], None, torch.from_numpy(frames).to(device), max_new=T + 1)
        errs, margins, below, flipped_below, wrong_above = [], [], 0, 0, 0
        for t in range(T):
            got = eng.last_logits().float().cpu()[0]
            r = trace[t]
            scale = r.abs().max().item()
            top2 = torch.topk(r, 2).values
            errs.append((got - r).abs().max().item() / scale)
            margins.append((top2[0] - top2[1]).item() / scale)
            same = int(got.argmax()) == gen[t]
            if above_id_line(r, budget):
                wrong_above += int(not same)
            else:
                below += 1
                flipped_below += int(not same)
            eng.set_current_tokens([gen[t]])
            eng.decode_step()
        e, m = np.array(errs), np.array(margins)
        return {"steps": T, "contexts": [768, 768 + T - 1], "id_line": 2 * budget, "rel_err_median": float(np.median(e)), "rel_err_p95": float(np.percentile(e, 95)),
                "rel_err_max": float(e.max()), "margin_median": float(np.median(m)), "margin_p05": float(np.percentile(m, 5)),
                "steps_below_the_id_line": below, "steps_actually_flipped": flipped_below + wrong_above, "wrong_above_the_line": wrong_above}

    out = {"what": "Emma-X-7B shape, 32 layers, RANDOM weights (seed 33), B=1 teacher-forced against the fp32 restatement (device-executed); errors and "
                   "margins relative to max|logit| of the step; id_line = 2 x the path's a-priori error budget (tests/conftest.py)"}
    model.engine.new_session(1, 512, 256 + 512 + T + 1)
    out["default_bf16_operands"] = run_mode(model, ID_BUDGET_FULL_DEPTH)
    model.engine.new_session(8, 512, 256 + 512 + 32)
    out["exact_numerics"] = run_mode(full_exact, ID_BUDGET_EXACT)
    print("\n512-step margin statistic:", json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_margin_statistic.json"), "w") as f:
        json.dump(out, f, indent=1)
    d, x = out["default_bf16_operands"], out["exact_numerics"]
    assert d["wrong_above_the_line"] == 0 and d["rel_err_max"] < TOL, d
    assert x["wrong_above_the_line"] == 0 and x["rel_err_max"] < TOL_EXACT and x["rel_err_median"] < TOL_EXACT / 2, x
    assert x["steps_actually_flipped"] <= 2, x


def test_exact_free_running_512_token_generations_equal_the_fp32_oracle(device, full, full_exact, dev_oracle):
    """FREE-RUNNING greedy decode at full depth on random weights (no teacher forcing, nothing planted): 512 new tokens of the exact
    session against the fp32 restatement's own greedy run (device-executed; tied to the CPU oracle by the test above), one frame + one
    512-token prompt per seed, four seeds.  VERDICT r05's mark: ids equal for >= 3 of 4 seeds.  Asserted: that, AND that any divergence
    starts on a step whose fp32 top-2 margin is below the exact path's id line (2e-4 of max|logit|: a genuine near-tie -- random weights
    put ~1 % of the steps there, and the two fp32 executions themselves differ by ~1e-5).  Results -> gpurun_out/r06_exact_free_running.json."""
    import json
    import os

    from conftest import ROOT

    cfg = full[0]
    ref = dev_oracle[0]
    T = 512
    rows_out, equal = [], 0
    for seed in (501, 502, 503, 504):
        rng = np.random.default_rng(seed)
        frames = rng.integers(0, 256, size=(1, 224, 224, 3), dtype=np.uint8)
        row = [1] + [int(x) for x in rng.integers(3, 31744, size=511)]
        gen, trace = _dev_trace(cfg, ref, ref, frames, row, T, device)
        ids, lens = full_exact.generate_ids([row], None, torch.from_numpy(frames).to(device), max_new_tokens=T, stop_on_eos=False)
        got = ids[0, : int(lens[0])].cpu().tolist()
        first = next((i for i, (a, b) in enumerate(zip(got, gen)) if a != b), None)
        rec = {"seed": seed, "tokens": T, "ids_equal": first is None, "first_divergence": first}
        if first is not None:
            top2 = torch.topk(trace[first], 2).values
            rec["fp32_margin_at_divergence"] = (top2[0] - top2[1]).item() / trace[first].abs().max().item()
            assert rec["fp32_margin_at_divergence"] < 2 * ID_BUDGET_EXACT, rec     # only a genuine near-tie may part the two runs
        equal += int(first is None)
        rows_out.append(rec)
    out = {"what": "Emma-X-7B shape, 32 layers, random weights (seed 33): free-running greedy generation of the exact-numerics session against the fp32 "
                   "restatement's greedy run, 512 new tokens, one frame + 512-token prompt per seed", "seeds_equal": equal, "runs": rows_out}
    print("\nexact free-running:", json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_exact_free_running.json"), "w") as f:
        json.dump(out, f, indent=1)
    assert equal >= 3, out


def test_exact_batch_8_rows_at_full_depth_equal_the_oracle_and_their_bs1_runs(device, full, dev_oracle):
    """SURVEY.md 0.4's criterion for the batched extension at configs[2]'s per-GPU shard (8 rows), full depth, RANDOM weights, FREE-RUNNING: every row of a
    ragged batch-8 generation of the exact-numerics session (decode_km.hip's EX kernels) emits the ids of the fp32 restatement's own bs = 1 greedy run, and
    rows re-run alone (bs = 1, decode_ks.hip's EX kernels) emit them too.  On the default bf16-operand path 3-4 of 8 rows keep their bs = 1 ids at this
    shape.  A divergence is tolerated only where the fp32 margin is below the exact path's id line (a genuine near-tie), and on at
    most one row.  Results -> gpurun_out/r06_exact_batch8_rows.json."""
    import json
    import os

    from conftest import ROOT
    from emmax.modeling import EmmaXForActionPrediction

    cfg, _, _, sd_bf = full
    ref = dev_oracle[0]
    T = 128
    model8 = EmmaXForActionPrediction(cfg, dict(sd_bf)).to(device, max_batch=8, max_prompt=512, max_ctx=256 + 512 + T + 8, exact=True)
    rng = np.random.default_rng(808)
    frames = rng.integers(0, 256, size=(8, 224, 224, 3), dtype=np.uint8)
    rows = [[1] + [int(x) for x in rng.integers(3, 31744, size=n - 1)] for n in (512, 300, 64, 17, 448, 129, 255, 512)]
    gens, traces = [], []
    for b in range(8):
        g, tr = _dev_trace(cfg, ref, ref, frames[b:b + 1], rows[b], T, device)
        gens.append(g)
        traces.append(tr)
    fr = torch.from_numpy(frames).to(device)
    ids8, _ = model8.generate_ids(rows, None, fr, max_new_tokens=T, stop_on_eos=False)
    recs, equal = [], 0

    def compare(got, b, what):
        want = gens[b]
        if cfg.eos_token_id in want:     # a row ends at its EOS (HF greedy semantics; emmax_generate pads behind it): compare up to and including it
            n = want.index(cfg.eos_token_id) + 1
            assert all(x == cfg.pad_token_id for x in got[n:]), (b, what)
            got, want = got[:n], want[:n]
        first = next((i for i, (x, y) in enumerate(zip(got, want)) if x != y), None)
        rec = {"row": b, "prompt_tokens": len(rows[b]), "run": what, "ids_equal": first is None, "first_divergence": first}
        if first is not None:
            top2 = torch.topk(traces[b][first], 2).values
            rec["fp32_margin_at_divergence"] = (top2[0] - top2[1]).item() / traces[b][first].abs().max().item()
            assert rec["fp32_margin_at_divergence"] < 2 * ID_BUDGET_EXACT, rec
        recs.append(rec)
        return first is None

    for b in range(8):
        equal += int(compare(ids8[b, :T].cpu().tolist(), b, "batch 8"))
    # four rows (two KV splits per head: the o-proj merges the fp32 partials; eight rows: one split, the attention launch writes the fp32 row itself)
    ids4, _ = model8.generate_ids(rows[1:5], None, fr[1:5], max_new_tokens=T, stop_on_eos=False)
    four = sum(int(compare(ids4[j, :T].cpu().tolist(), b, "batch 4")) for j, b in enumerate(range(1, 5)))
    alone = 0
    for b in (0, 3, 6):
        ids1, _ = model8.generate_ids([rows[b]], None, fr[b:b + 1], max_new_tokens=T, stop_on_eos=False)
        alone += int(compare(ids1[0, :T].cpu().tolist(), b, "bs 1"))
    out = {"what": "Emma-X-7B shape, 32 layers, random weights (seed 33), exact numerics: FREE-RUNNING ragged batch-8 generation (prompts of 17..512 tokens, %d new tokens) "
                   "against the fp32 restatement's bs = 1 greedy run of every row; three rows re-run alone" % T,
           "rows_equal_in_the_batch": equal, "rows_equal_in_a_batch_of_four": four, "rows_equal_alone": alone, "runs": recs}
    print("\nexact batch-8 rows:", json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_exact_batch8_rows.json"), "w") as f:
        json.dump(out, f, indent=1)
    del model8
    torch.cuda.empty_cache()
    assert equal >= 7 and four >= 3 and alone >= 2, out
