"""
bench.py -- headline measurement of the Emma-X hot path on MI355X.

One "step" = one generate_actions pass over one batch of synthetic frames on every rank: fused DINOv2+SigLIP encode ->
projector -> LLaMA-2-7B prefill (256 patches + 512 prompt tokens) -> 512 greedy decode steps (EOS disabled so every step
does the full work) -> action de-tokenisation (+ one RCCL all_gather of the results when N > 1).
Workload at N=1 = BASELINE.json configs[1] (one frame); N>1 = configs[2]'s shard on every rank: 8 frames per GPU (64 frames over
8 GPUs), weak scaling, data parallel, the line then carries `rccl_ranks` (counted by a real collective), `gather_ms` and its share.

Prints ONE JSON line (rank 0) with the driver's contract + `roofline` (dominant kernel = gate/up decode GEMV, timed
live with HIP events through emmax_profile_decode_stage) + `cpu_baseline` (the oracle timed on the host cores, N=1 only).
"""

import argparse
import json
import os
import sys
import time

# hipGraph replay (--graph): ROCm 7.2's default replay path ("graph packet capture") costs ~0.65 us per kernel node on the device -- 4 % of a
# 163-node decode step; with the runtime switch below the replay runs at the eager rate (profiles/r05_graph_switches.txt: 2.5745 against 2.5749
# ms/token eager and 2.69 default replay).  The runtime reads it once, at its first call: set before anything touches HIP (emmax/__init__.py does
# the same when EMMAX_GRAPH=1 is exported)
if "--graph" in sys.argv:
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

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


def cpu_baseline(cfg, prompt_tokens, new_tokens, seed=0, decode_steps=16, distinct_layers=4):
    """Time the CPU oracle (fp32, host cores) on a bounded sample of the same workload.

    Executed in full, once: both ViT towers (every block up to take_index) + the projector on one frame, and the LLaMA prefill
    at S = 256 + P over all `num_layers` layer passes + the lm-head.  Then `decode_steps` cached greedy steps (every layer pass,
    growing KV cache, lm-head + argmax each) are timed and their mean is extrapolated to the remaining new tokens:
        t_action = t_vision + t_projector + t_prefill + t_head + (T - 1) * mean(t_decode_step).
    The only economy is the weight VALUES: `distinct_layers` random LLaMA layers are cycled through the layer passes (fp32 they
    are 0.8 GB each -- four of them are far larger than any host cache, so every pass streams from DRAM like 32 distinct
    layers would) instead of generating 27 GB of random numbers.  PyTorch CPU ops do not scale to every hardware thread of a
    big host: the thread count is chosen from {32, 64, all} by timing a decode layer pass and reported as `cores`."""
    from emmax.weights import tower_param_shapes
    from oracle import emmax_oracle as orc

    torch.manual_seed(seed)
    ncpu = os.cpu_count() or 1
    L = cfg.llm
    S = cfg.n_patches + prompt_tokens
    nl = L.num_layers
    nd = max(1, min(distinct_layers, nl))

    def w(*shape):
        return torch.randn(*shape) * 0.02

    qd, kvd = L.num_heads * L.head_dim, L.num_kv_heads * L.head_dim
    sd = {}
    for i in range(nd):
        p = f"language_model.model.layers.{i}."
        sd.update({p + "input_layernorm.weight": torch.ones(L.hidden_size), p + "post_attention_layernorm.weight": torch.ones(L.hidden_size),
                   p + "self_attn.q_proj.weight": w(qd, L.hidden_size), p + "self_attn.k_proj.weight": w(kvd, L.hidden_size),
                   p + "self_attn.v_proj.weight": w(kvd, L.hidden_size), p + "self_attn.o_proj.weight": w(L.hidden_size, qd),
                   p + "mlp.gate_proj.weight": w(L.intermediate_size, L.hidden_size), p + "mlp.up_proj.weight": w(L.intermediate_size, L.hidden_size),
                   p + "mlp.down_proj.weight": w(L.hidden_size, L.intermediate_size)})
    head = w(L.vocab_size, L.hidden_size)
    norm_w = torch.ones(L.hidden_size)
    t = {}
    with torch.inference_mode():
        # ---- thread count: one cached decode layer pass at context S (the dominant term) ----
        h0 = torch.randn(1, S, L.hidden_size)
        x0 = torch.randn(1, 1, L.hidden_size)
        torch.set_num_threads(min(ncpu, 64))
        _, kv0 = orc.llama_layer(h0, sd, 0, L, torch.arange(S), None, torch.float32)
        probe = lambda: orc.llama_layer(x0, sd, 0, L, torch.arange(S, S + 1), kv0, torch.float32)
        best = None
        for nt in sorted({min(ncpu, 32), min(ncpu, 64), ncpu}):
            torch.set_num_threads(nt)
            dt = _timed(probe, reps=3)
            if best is None or dt < best[1]:
                best = (nt, dt)
        nthreads = best[0]
        torch.set_num_threads(nthreads)
        del kv0
        # ---- vision + projector, in full, one frame (second call timed: the first creates the oneDNN primitives) ----
        pix = torch.randn(1, 6, 224, 224)
        sdv = {}
        for i, tw in enumerate(cfg.towers):
            for k, shp, _ in tower_param_shapes(tw):
                blk = int(k.split(".")[1]) if k.startswith("blocks.") else -1
                if blk <= tw.take_index:
                    sdv[orc.TOWER_PREFIXES[i] + k] = torch.randn(*shp) * 0.02
        v, p1, hdim, _ = cfg.projector_dims
        sdv.update({"projector.fc1.weight": w(p1, v), "projector.fc1.bias": torch.zeros(p1), "projector.fc2.weight": w(hdim, p1),
                    "projector.fc2.bias": torch.zeros(hdim), "projector.fc3.weight": w(hdim, hdim), "projector.fc3.bias": torch.zeros(hdim)})
        feats = {}
        t["vision_towers"] = _timed(lambda: feats.__setitem__("f", orc.vision_backbone(pix, sdv, cfg)))
        t["projector"] = _timed(lambda: orc.projector(feats["f"], sdv))
        del sdv
        # ---- prefill: all layer passes at S rows, once (the thread-count probe above warmed the primitives up) ----
        caches = []
        t0 = time.perf_counter()
        h = h0
        pos = torch.arange(S)
        for li in range(nl):
            h, kv = orc.llama_layer(h, sd, li % nd, L, pos, None, torch.float32)
            caches.append(kv)
        t["prefill_layers"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        nxt = int(torch.argmax(torch.nn.functional.linear(orc.rms_norm(h[:, -1:], norm_w, L.rms_eps), head)))
        t["lm_head"] = time.perf_counter() - t0
        # ---- cached greedy decode: `decode_steps` full steps, KV cache growing ----
        steps = []
        x = x0
        for st in range(decode_steps):
            t0 = time.perf_counter()
            hh = x
            pos = torch.arange(S + st, S + st + 1)
            for li in range(nl):
                hh, caches[li] = orc.llama_layer(hh, sd, li % nd, L, pos, caches[li], torch.float32)
            nxt = int(torch.argmax(torch.nn.functional.linear(orc.rms_norm(hh, norm_w, L.rms_eps), head)))
            steps.append(time.perf_counter() - t0)
            x = torch.roll(x0, nxt % 97, dims=-1)   # next-token embedding stand-in (the 8 KB row gather is not timed work)
        t["decode_step_mean"] = float(np.mean(steps))
        t["decode_step_min"] = float(np.min(steps))
    t_action = t["vision_towers"] + t["projector"] + t["prefill_layers"] + t["lm_head"] + (new_tokens - 1) * t["decode_step_mean"]
    return {"value": 1.0 / t_action, "unit": "actions/s", "cores": nthreads, "kind": "port",
            "sample": ("fp32 PyTorch oracle on %d of %d host threads: both ViT towers (%d+%d blocks) + projector on one frame and the "
                       "LLaMA prefill (S=%d, all %d layer passes) + lm-head executed in full, once; then %d cached greedy decode "
                       "steps (all %d layer passes + lm-head + argmax each, context %d..%d) timed, their mean extrapolated to the "
                       "other %d new tokens; %d distinct random layers (%.1f GB fp32) are cycled through the layer passes" %
                       (nthreads, ncpu, cfg.towers[0].take_index + 1, cfg.towers[1].take_index + 1, S, nl, decode_steps, nl, S, S + decode_steps - 1,
                        new_tokens - 1 - decode_steps, nd, nd * (4 * qd * L.hidden_size + 3 * L.intermediate_size * L.hidden_size) * 4 / 1e9)),
            "seconds_per_action": round(t_action, 3), "parts_s": {k: round(v, 5) for k, v in t.items()}}


def _alone_baseline(args, model, frames, prompts, T, rank, world, B):
    """N > 1: the one-GPU rate of the SAME per-GPU workload, measured live in this very job -- rank 0 runs `--alone-steps` steps of
    its own shard while every other rank is parked at the barrier below (nothing else touches rank 0's GPU; there is no
    collective inside a step anyway).  `--scale-baseline V` replaces the measurement.  Runs before the warm-up, outside the
    timed region."""
    from emmax import dist as edist

    if world == 1:
        return None
    if args.scale_baseline is not None:
        return {"value": float(args.scale_baseline), "unit": "actions/s", "source": "--scale-baseline (given on the command line)"}
    val = None
    if rank == 0:
        model.generate_actions_batch(frames, prompts, max_new_tokens=T, stop_on_eos=False)   # warm-up (allocations, first launches)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(1, args.alone_steps)):
            model.generate_actions_batch(frames, prompts, max_new_tokens=T, stop_on_eos=False)
        torch.cuda.synchronize()
        val = B * max(1, args.alone_steps) / (time.perf_counter() - t0)
    edist.barrier()
    if val is None:
        return None
    return {"value": round(val, 4), "unit": "actions/s",
            "source": "live: rank 0 alone on its GPU, %d step(s) of the same %d-frame shard, other ranks parked at a barrier" % (max(1, args.alone_steps), B)}


def _self_launch(n):
    """Re-exec this script under `python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1` with the
    same arguments (an external torchrun sets WORLD_SIZE and never gets here)."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL needs it)
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, argv)


def _workload(args, world, B, P, T):
    if args.tiny:
        return "TINY plumbing config (invalid as headline)"
    if args.fp8:
        head = "BASELINE configs[4] (fp8-e4m3 decode weights, batch %d%s)" % (B, ", hipGraph step" if args.graph else "")
    elif world == 1 and B == 1:
        head = "BASELINE configs[1]"
    elif B == 8:
        head = "BASELINE configs[2] shard (8 frames/GPU; 64 frames over 8 GPUs at N=8), data-parallel over %d GPU(s), one RCCL all_gather of the results per batch" % world
    else:
        head = "BASELINE configs[1] extended to %d frame(s)/GPU" % B
    if getattr(args, "kv_fp8", False):
        head += " [opt-in fp8-e4m3 KV cache: NOT the bf16 headline numerics]"
    if getattr(args, "exact", False):
        head += " [EXACT NUMERICS: fp32 activations, two-term bf16 MFMA / dot2 operands, fp32 attention over a %s KV cache -- the reference's fp32 CPU arithmetic]" % ("fp32" if getattr(args, "exact_fp32kv", False) else "24-bit") + ""
    return head + (": Emma-X-7B bf16, %d frame(s)/GPU 224x224, %d-token prompt, greedy, %d new tokens (EOS disabled), random-init weights; "
                   "uint8 frames and prompt ids are resident in HBM when the timed region starts; the 7-vector is de-tokenised from the "
                   "generated ids by the ids-level stand-in `actions_from_ids` (no LLaMA tokenizer offline: decode -> Solver text parse is "
                   "host work of microseconds, covered by tests, not in the timed step)" % (B, P, T))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-per-gpu", type=int, default=None,
                    help="frames per GPU per step; default 1 at --gpus 1 (BASELINE configs[1]), 8 at --gpus > 1 (configs[2]: 64 frames over 8 GPUs)")
    ap.add_argument("--prompt-tokens", type=int, default=512)
    ap.add_argument("--new-tokens", type=int, default=512)
    ap.add_argument("--tiny", action="store_true", help="tiny config (plumbing check only; NOT a valid headline number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the decode step as a hipGraph (EMMAX_GRAPH=1) instead of eager launch-ahead")
    ap.add_argument("--fp8", action="store_true", help="BASELINE config 5: fp8-e4m3 decode weights (NOT the bf16 headline)")
    ap.add_argument("--kv-fp8", action="store_true", help="opt-in fp8-e4m3 KV cache, one scale per (token, head) row (NOT the bf16 headline)")
    ap.add_argument("--exact", action="store_true", help="exact numerics (tuning switch exact): fp32 activations, two-term bf16 operands, fp32 KV cache -- "
                                                         "the reference's fp32 CPU arithmetic; batch 1-2 (a conformance mode: the line says so)")
    ap.add_argument("--exact-fp32kv", action="store_true", help="--exact with an fp32 K / V cache (tuning switch exact = 2) instead of the 24-bit one: the A/B partner")
    ap.add_argument("--scale-baseline", type=float, default=None,
                    help="N > 1 only: actions/s of ONE GPU on the same per-GPU workload, measured elsewhere; default: rank 0 measures it "
                         "live (alone on its GPU, the other ranks parked at a barrier) before the group run")
    ap.add_argument("--alone-steps", type=int, default=2, help="N > 1 only: steps of the live one-GPU baseline run on rank 0")
    args = ap.parse_args()
    if args.kv_fp8:
        os.environ["EMMAX_KV_FP8"] = "1"   # (as EMMAX_GRAPH: read once at library start-up, inherited by self-launched ranks)
    if args.exact_fp32kv:
        args.exact = True
    if args.exact:
        os.environ["EMMAX_EXACT"] = "2" if args.exact_fp32kv else "1"
    if args.graph:
        os.environ["EMMAX_GRAPH"] = "1"   # read once by the library at start-up (include/emmax.h: tuning switches); inherited by self-launched ranks
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the driver's own launch line (one process per GPU, RCCL)
        _self_launch(args.gpus)

    from emmax import dist as edist
    from emmax.config import EmmaXConfig
    from emmax.modeling import EmmaXForActionPrediction

    if args.batch_per_gpu is None:
        args.batch_per_gpu = 1 if args.gpus == 1 else 8
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
    # EOS disabled: no token id equals -1, so no row can end early on these random weights and every one of the T decode steps does its full work
    # (a finished row would skip its K / V reads); checked after the timed loop -- every row must have emitted all T tokens
    cfg.eos_token_id = -1
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

    alone = _alone_baseline(args, model, frames, prompts, T, rank, world, B)
    gather_s = []

    last_lens = []

    def step():
        acts, ids, lens = model.generate_actions_batch(frames, prompts, max_new_tokens=T, stop_on_eos=False)
        last_lens[:] = [lens]
        a = torch.from_numpy(acts).to(dev)
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        out = edist.gather_results(a, ids, lens)      # the ONE collective of the data path (RCCL all_gather over xGMI when N > 1)
        torch.cuda.synchronize()
        gather_s.append(time.perf_counter() - g0)
        return out

    for _ in range(args.warmup):
        step()
    gather_s.clear()
    rccl_ranks = edist.collective_world_size(dev)     # counted by an actual all_reduce, not read from the environment
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
    gather_ms = edist.max_over_ranks(float(np.mean(gather_s)) * 1e3 if gather_s else 0.0, dev)
    if last_lens and int(last_lens[0].min()) != T:
        raise SystemExit("bench: a row stopped after %d of %d tokens -- the timed steps did not do the full work" % (int(last_lens[0].min()), T))

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
            "paged_attn": B * 2 * ctx * kvd * (1 if args.kv_fp8 else (4 if args.exact_fp32kv else 3) if args.exact else 2) + (B * 2 * ctx * L.num_kv_heads * 4 if args.kv_fp8 else 0),
            "oproj_gemv": L.hidden_size * qd * wbytes,
            "gateup_gemv": 2 * inter_p * L.hidden_size * wbytes,
            "down_gemv": L.hidden_size * inter_p * wbytes,
            "lmhead_argmax": L.vocab_size * L.hidden_size * wbytes,
        }
        dom = "gateup_gemv"
        achieved = stage_bytes[dom] / (stage_us[dom] * 1e-6) / 1e9
        # HBM traffic of the same kernel from PMC counters (separate rocprofv3 --pmc passes over tools/pmc_probe.py with the same
        # batch; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md): collected offline by tools/profile_round.sh on
        # the kernels this line names, not during this run -- null when no pass exists for this (batch, weight format), and on every
        # N > 1 line (it carries no number that was not measured in its own run)
        traffic = None
        pmc_name = "r06_pmc_traffic_b%d.json" % B if (world == 1 and not args.fp8 and not args.kv_fp8 and not args.exact) else None
        pmc_file = os.path.join(ROOT, "profiles", pmc_name or "none")
        if not args.tiny and pmc_name and os.path.isfile(pmc_file):
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
        kv_bytes = L.num_layers * B * 2 * (ctx + nsteps // 2) * (kvd * (1 if args.kv_fp8 else (4 if args.exact_fp32kv else 3) if args.exact else 2) + (L.num_kv_heads * 4 if args.kv_fp8 else 0))
        step_gbs = (w_llm + kv_bytes) / (step_ms * 1e-3) / 1e9
        out = {
            "metric": "actions/sec", "value": round(actions_per_s, 4), "unit": "actions/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": ("bf16 weights x two-term bf16 activations (fp32-equivalent), fp32 accumulate" if args.exact else "bf16") if not args.fp8 else "bf16 activations / fp8-e4m3 decode weights", "data": "synthetic",
            "cmd": "python bench.py --gpus %d --steps %d --warmup %d --batch-per-gpu %d --prompt-tokens %d --new-tokens %d%s%s%s"
                   % (world, args.steps, args.warmup, B, P, T, " --fp8" if args.fp8 else "", (" --graph" if args.graph else "") + (" --kv-fp8" if args.kv_fp8 else "") + (" --exact-fp32kv" if args.exact_fp32kv else " --exact" if args.exact else ""), " --tiny" if args.tiny else ""),
            "config": {"workload": _workload(args, world, B, P, T),
                       "batch_per_gpu": B, "global_batch": B * world, "prompt_tokens": P, "new_tokens": T, "context": ctx + T,
                       "parallelism": f"dp{world}", "hipgraph": eng.graph_active(), "exact_numerics": bool(eng.exact)},
            "value_per_gpu": round(actions_per_s / world, 4), "single_gpu_same_workload": alone,
            "scaling_efficiency": round(actions_per_s / (world * alone["value"]), 4) if alone else None,
            "rccl_ranks": rccl_ranks, "dist_backend": edist.backend_name(),
            "gather_ms": round(gather_ms, 4), "gather_share": round(gather_ms / ms_per_step, 6),
            "p50_latency_ms": round(float(np.median(lat)) * 1e3, 2),
            "decode_ms_per_token": round(step_ms, 4), "decode_tokens_per_s": round(B * 1e3 / step_ms, 1),
            "decode_step_hbm_gbs": round(step_gbs, 1), "decode_step_hbm_frac": round(step_gbs / HBM_PEAK_GBS, 4),
            "weights": {"arena_gb": round(eng.arena.numel() / 1e9, 2), "aux_gb_batch_ge_3": round((eng.weight_bytes() - eng.arena.numel()) / 1e9, 2),
                        "finalize_s": round(eng.finalize_s, 3), "aux_build_s": round(eng.aux_build_s, 3)},
            "stage_us": {k: round(v, 2) for k, v in stage_us.items()},
            "stage_gbs": {k: round(stage_bytes[k] / (stage_us[k] * 1e-6) / 1e9, 1) for k in stage_names},
            "roofline": {"kernel": ("emmax_decode_ks_kernel<B=%d,GATEUP,NORM,CPL=1%s>" % (B, ",EX" if args.exact else "") if B <= 2 and not args.fp8 else ("emmax_decode_kmp_kernel<GATEUP,NORM,TMAX=6%s> (B=%d)" % (",NH=2" if B > 32 else "", B) if B > 16 else "emmax_decode_km_kernel<GATEUP,NORM%s> (B=%d)" % (",FP8" if args.fp8 else ",EX" if args.exact else "", B))) + " (gate/up projection + SiLU*mul)", "bound": "hbm",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "bytes_per_launch": stage_bytes[dom], "us_per_launch": round(stage_us[dom], 2), "traffic": traffic,
                         "traffic_source": "profiles/%s (rocprofv3 --pmc, offline)" % pmc_name if traffic else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, P, T)
        print(json.dumps(out), flush=True)
    edist.barrier()


if __name__ == "__main__":
    main()
