"""
bench.py -- headline measurement of the Emma-X hot path on MI355X.

One "step" = one generate_actions pass over one batch of synthetic frames on every rank: fused DINOv2+SigLIP encode ->
projector -> LLaMA-2-7B prefill (256 patches + 512 prompt tokens) -> 512 greedy decode steps (EOS disabled so every step
does the full work) -> action de-tokenisation (+ one RCCL all_gather of the results when N > 1).
Workload at N=1 = BASELINE.json configs[1]; N>1 = the same per-GPU work on every rank (weak scaling, data parallel).

Prints ONE JSON line (rank 0) with the driver's contract + `roofline` (dominant kernel = gate/up decode GEMV, timed
live with HIP events through emmax_profile_decode_stage) + `cpu_baseline` (the oracle timed on the host cores, N=1 only).
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "emma-x_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def _timed(fn, reps=1):
    """Run once untimed (oneDNN primitive creation, thread-pool spin-up), then `reps` timed calls; seconds per call."""
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def cpu_baseline(cfg, prompt_tokens, new_tokens, seed=0):
    """Time the CPU oracle (fp32, host cores) on a bounded sample of the same workload and extrapolate linearly.

    Sample: ONE full-size LLaMA decoder layer (prefill at S = 256+P, cached decode steps at that context), the lm-head,
    ONE block (+ patch embed) of each ViT tower on one frame, and the projector.  Everything else is the same layer
    repeated, so
        t_action = sum_t (take_t+1)*t_block_t + t_proj + 32*t_prefill_layer + t_head + (T-1) * (32*t_decode_layer + t_head).
    PyTorch CPU ops do not scale to every hardware thread of a big host: the thread count is chosen from {32, 64, all}
    by timing the decode layer (the dominant term) and is reported as `cores`."""
    from emmax.weights import tower_param_shapes
    from oracle import emmax_oracle as orc

    torch.manual_seed(seed)
    ncpu = os.cpu_count() or 1
    L = cfg.llm
    S = cfg.n_patches + prompt_tokens

    def w(*shape):
        return torch.randn(*shape) * 0.02

    p = "language_model.model.layers.0."
    qd, kvd = L.num_heads * L.head_dim, L.num_kv_heads * L.head_dim
    sd = {p + "input_layernorm.weight": torch.ones(L.hidden_size), p + "post_attention_layernorm.weight": torch.ones(L.hidden_size),
          p + "self_attn.q_proj.weight": w(qd, L.hidden_size), p + "self_attn.k_proj.weight": w(kvd, L.hidden_size),
          p + "self_attn.v_proj.weight": w(kvd, L.hidden_size), p + "self_attn.o_proj.weight": w(L.hidden_size, qd),
          p + "mlp.gate_proj.weight": w(L.intermediate_size, L.hidden_size), p + "mlp.up_proj.weight": w(L.intermediate_size, L.hidden_size),
          p + "mlp.down_proj.weight": w(L.hidden_size, L.intermediate_size)}
    head = w(L.vocab_size, L.hidden_size)
    t = {}
    with torch.inference_mode():
        h = torch.randn(1, S, L.hidden_size)
        x = torch.randn(1, 1, L.hidden_size)
        torch.set_num_threads(min(ncpu, 64))
        _, kv = orc.llama_layer(h, sd, 0, L, torch.arange(S), None, torch.float32)
        dec = lambda: orc.llama_layer(x, sd, 0, L, torch.arange(S, S + 1), kv, torch.float32)
        best = None
        for nt in sorted({min(ncpu, 32), min(ncpu, 64), ncpu}):
            torch.set_num_threads(nt)
            dt = _timed(dec, reps=3)
            if best is None or dt < best[1]:
                best = (nt, dt)
        nthreads = best[0]
        torch.set_num_threads(nthreads)
        t["decode_layer"] = best[1]
        t["prefill_layer"] = _timed(lambda: orc.llama_layer(h, sd, 0, L, torch.arange(S), None, torch.float32))
        t["lm_head"] = _timed(lambda: torch.nn.functional.linear(orc.rms_norm(x, torch.ones(L.hidden_size), L.rms_eps), head), reps=3)
        del sd, kv, head
        pix = torch.randn(1, 6, 224, 224)
        vt = 0.0
        for i, tw in enumerate(cfg.towers):
            sdv = {orc.TOWER_PREFIXES[i] + k: torch.randn(*shp) * 0.02 for k, shp, _ in tower_param_shapes(tw)
                   if k.startswith("blocks.0.") or not k.startswith("blocks.")}
            vt += (tw.take_index + 1) * _timed(lambda: orc.vit_tower(pix[:, 3 * i:3 * i + 3], sdv, orc.TOWER_PREFIXES[i], tw,
                                                                     torch.float32, n_blocks=1))
        t["vision_towers"] = vt
        v, p1, hdim, _ = cfg.projector_dims
        sdp = {"projector.fc1.weight": w(p1, v), "projector.fc1.bias": torch.zeros(p1), "projector.fc2.weight": w(hdim, p1),
               "projector.fc2.bias": torch.zeros(hdim), "projector.fc3.weight": w(hdim, hdim), "projector.fc3.bias": torch.zeros(hdim)}
        feats = torch.randn(1, cfg.n_patches, v)
        t["projector"] = _timed(lambda: orc.projector(feats, sdp))
    nl = L.num_layers
    t_action = (t["vision_towers"] + t["projector"] + nl * t["prefill_layer"] + t["lm_head"]
                + (new_tokens - 1) * (nl * t["decode_layer"] + t["lm_head"]))
    return {"value": 1.0 / t_action, "unit": "actions/s", "cores": nthreads, "kind": "port",
            "sample": ("fp32 PyTorch oracle on %d of %d host threads: 1 of %d LLaMA layers (prefill S=%d, cached decode at that context), "
                       "lm-head, 1 block + patch-embed per ViT tower, projector; each timed after one untimed call and extrapolated "
                       "linearly to %d layers / %d+%d blocks / %d new tokens" %
                       (nthreads, ncpu, nl, S, nl, cfg.towers[0].take_index + 1, cfg.towers[1].take_index + 1, new_tokens)),
            "seconds_per_action": round(t_action, 3), "parts_s": {k: round(v, 5) for k, v in t.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-per-gpu", type=int, default=1)
    ap.add_argument("--prompt-tokens", type=int, default=512)
    ap.add_argument("--new-tokens", type=int, default=512)
    ap.add_argument("--tiny", action="store_true", help="tiny config (plumbing check only; NOT a valid headline number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the decode step as a hipGraph (EMMAX_GRAPH=1) instead of eager launch-ahead")
    ap.add_argument("--fp8", action="store_true", help="BASELINE config 5: fp8-e4m3 decode weights (NOT the bf16 headline)")
    args = ap.parse_args()
    if args.graph:
        os.environ["EMMAX_GRAPH"] = "1"

    from emmax import dist as edist
    from emmax.config import EmmaXConfig
    from emmax.modeling import EmmaXForActionPrediction

    rank, world, local = edist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    if os.environ.get("EMMAX_FORCE_DEVICE") is not None:   # test hook: several ranks on one GPU (with EMMAX_DIST_BACKEND=gloo)
        local = int(os.environ["EMMAX_FORCE_DEVICE"])
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    cfg = EmmaXConfig.tiny() if args.tiny else EmmaXConfig.emma_x_7b()
    if args.fp8:
        cfg.decode_weight_dtype = "fp8"
    wbytes = 1 if args.fp8 else 2   # bytes per decode weight
    B, P, T = args.batch_per_gpu, args.prompt_tokens, args.new_tokens
    model = EmmaXForActionPrediction.from_synthetic(cfg, seed=0, device=dev, max_batch=B, max_prompt=P,
                                                    max_ctx=cfg.n_patches + P + T + 1)
    # synthetic inputs: frames U{0..255}, prompts [BOS] + U{3..31743}; every rank gets its own shard
    rng = np.random.default_rng(1234 + rank)
    frames = torch.from_numpy(rng.integers(0, 256, size=(B, 224, 224, 3), dtype=np.uint8)).to(dev)
    prompts = [[1] + [int(x) for x in rng.integers(3, 31744, size=P - 1)] for _ in range(B)]

    def step():
        acts, ids, lens = model.generate_actions_batch(frames, prompts, max_new_tokens=T, stop_on_eos=False)
        return edist.gather_results(torch.from_numpy(acts).to(dev), ids, lens)

    for _ in range(args.warmup):
        step()
    edist.barrier()
    torch.cuda.synchronize()
    lat = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - s0)
    edist.barrier()
    torch.cuda.synchronize()
    elapsed = edist.max_over_ranks(time.perf_counter() - t0, dev)
    ms_per_step = elapsed / args.steps * 1e3
    actions_per_s = world * B * args.steps / elapsed

    # ---- stage breakdown + roofline of the dominant kernel (rank 0) ----
    out = None
    if rank == 0:
        eng = model.engine
        L = cfg.llm
        model._prefill(prompts, None, frames, max_new=T)   # active session at context 256+P
        stage_names = ["qkv_gemv", "paged_attn", "oproj_gemv", "gateup_gemv", "down_gemv", "lmhead_argmax"]
        stage_us = {n: eng.profile_decode_stage(i, reps=3) for i, n in enumerate(stage_names)}
        qd, kvd = L.num_heads * L.head_dim, L.num_kv_heads * L.head_dim
        inter_p = (L.intermediate_size + 63) // 64 * 64
        ctx = cfg.n_patches + P
        stage_bytes = {
            "qkv_gemv": (qd + 2 * kvd) * L.hidden_size * wbytes,
            "paged_attn": B * 2 * ctx * kvd * 2,
            "oproj_gemv": L.hidden_size * qd * wbytes,
            "gateup_gemv": 2 * inter_p * L.hidden_size * wbytes,
            "down_gemv": L.hidden_size * inter_p * wbytes,
            "lmhead_argmax": L.vocab_size * L.hidden_size * wbytes,
        }
        dom = "gateup_gemv"
        achieved = stage_bytes[dom] / (stage_us[dom] * 1e-6) / 1e9
        # HBM traffic of the same kernel from PMC counters (separate rocprofv3 --pmc passes over tools/pmc_probe.py, B=1;
        # FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md); not collected during this run
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if B == 1 and not args.tiny and not args.fp8 and os.path.isfile(pmc_file):
            with open(pmc_file) as f:
                traffic = json.load(f)["stages"].get(dom, {}).get("hbm_bytes_per_launch")
        # whole decode step: algorithmic bytes / measured step time
        t_s = time.perf_counter()
        nsteps = 64
        _, _ = eng.generate(nsteps + 1, False)
        torch.cuda.synchronize()
        step_ms = (time.perf_counter() - t_s) / nsteps * 1e3
        w_llm = (L.num_layers * ((qd + 2 * kvd) * L.hidden_size + L.hidden_size * qd + 3 * L.intermediate_size * L.hidden_size
                                 + 2 * L.hidden_size) + L.hidden_size + L.vocab_size * L.hidden_size) * wbytes
        kv_bytes = L.num_layers * B * 2 * (ctx + nsteps // 2) * kvd * 2
        step_gbs = (w_llm + kv_bytes) / (step_ms * 1e-3) / 1e9
        out = {
            "metric": "actions/sec", "value": round(actions_per_s, 4), "unit": "actions/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if not args.fp8 else "bf16 activations / fp8-e4m3 decode weights", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[4] (fp8 decode weights): " if args.fp8 else "") + ("BASELINE configs[1]: Emma-X-7B bf16, %d frame(s)/GPU 224x224, %d-token prompt, greedy, %d new tokens "
                                    "(EOS disabled), random-init weights" % (B, P, T)) if not args.tiny else "TINY plumbing config (invalid as headline)",
                       "batch_per_gpu": B, "global_batch": B * world, "prompt_tokens": P, "new_tokens": T, "context": ctx + T,
                       "parallelism": f"dp{world}", "hipgraph": eng.graph_active(), "chained_launch": eng.chain_active()},
            "p50_latency_ms": round(float(np.median(lat)) * 1e3, 2),
            "decode_ms_per_token": round(step_ms, 4), "decode_tokens_per_s": round(B * 1e3 / step_ms, 1),
            "decode_step_hbm_gbs": round(step_gbs, 1), "decode_step_hbm_frac": round(step_gbs / HBM_PEAK_GBS, 4),
            "stage_us": {k: round(v, 2) for k, v in stage_us.items()},
            "stage_gbs": {k: round(stage_bytes[k] / (stage_us[k] * 1e-6) / 1e9, 1) for k in stage_names},
            "roofline": {"kernel": ("emmax_decode_gemv_kernel<B=%d,GATEUP,NORM>" % B if (B <= 2 and not args.fp8) else "emmax_decode_mfma_kernel<GATEUP,NORM%s> (B=%d)" % (",FP8" if args.fp8 else "", B)) + " (gate/up projection + SiLU*mul)", "bound": "hbm",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "bytes_per_launch": stage_bytes[dom], "us_per_launch": round(stage_us[dom], 2), "traffic": traffic,
                         "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc, offline)" if traffic else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, P, T)
        print(json.dumps(out), flush=True)
    edist.barrier()


if __name__ == "__main__":
    main()
