"""Which bf16 rounding point carries the logit error of the default path?  ("first an ABLATION with one point
switched at a time").  TEST / MEASUREMENT INFRASTRUCTURE, not product code: the fp32 restatement of oracle/emmax_oracle.py, executed by
torch on the GPU at the full Emma-X-7B shape (random weights, seed 33 -- the weights of tests/test_full_depth_gpu.py), with ONE named
activation rounded to bf16 the way the HIP path's default mode rounds it, everything else fp32; 64 teacher-forced decode steps after the
768-row prefill; per configuration the median / max of |logit - fp32 logit| / max|fp32 logit|.

  points   xn       the normalised rows that enter qkv / gate-up / lm-head (RMSNorm output)
           q        the rotated queries          kv    the K / V rows kept in the cache (and read by the prefill attention)
           p        the softmax probabilities (the prefill's PV MFMA operand; the decode attention keeps them fp32)
           attn     the attention output that enters the o-proj
           act      the SwiGLU product that enters the down projection
           resid    the residual stream itself (the bf16 rows of rounds 1-4; fp32 since round 5)
           vit      every GEMM / attention operand inside the two towers and the projector (LayerNorm output, q / k / v, P, attention
                    output, GELU output, features), and the patch embeddings handed to the LLM
  rows     each point alone ("only"), everything BUT that point ("all_but"), all of them ("all" = an emulation of the default path), none.

Writes gpurun_out/r06_exact_ablation.json (committed as profiles/r06_exact_ablation.json).  Next to it the MEASURED lines of the real
kernels come from tests/test_full_depth_gpu.py (r06_margin_statistic.json: default path 2.4e-2, exact numerics ~1e-5)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
import numpy as np
import torch
import torch.nn.functional as F

from oracle import emmax_oracle as orc

BF = torch.bfloat16
POINTS = ("xn", "q", "kv", "p", "attn", "act", "resid", "vit")


def r(x, on):
    return x.to(BF).float() if on else x


def vit_tower(x, sd, prefix, tw, rv):
    g = lambda k: sd[prefix + k]
    B = x.shape[0]
    D, H = tw.embed_dim, tw.num_heads
    hd = D // H
    t = F.conv2d(r(x, rv), g("patch_embed.proj.weight"), g("patch_embed.proj.bias"), stride=tw.patch).flatten(2).transpose(1, 2)
    t = r(t, rv) + g("pos_embed")
    if tw.n_prefix > 0:
        pre = [g("cls_token").expand(B, -1, -1)]
        if tw.n_reg > 0:
            pre.append(g("reg_token").expand(B, -1, -1))
        t = torch.cat(pre + [t], dim=1)
    t = r(t, rv)
    for i in range(tw.take_index + 1):
        p = f"blocks.{i}."
        h = r(F.layer_norm(t, (D,), g(p + "norm1.weight"), g(p + "norm1.bias"), eps=tw.ln_eps), rv)
        qkv = r(F.linear(h, g(p + "attn.qkv.weight"), g(p + "attn.qkv.bias")), rv)
        N = qkv.shape[1]
        q, k, v = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4).unbind(0)
        att = r(torch.matmul(q * (hd ** -0.5), k.transpose(-2, -1)).softmax(dim=-1), rv)
        a = r(torch.matmul(att, v).transpose(1, 2).reshape(B, N, D), rv)
        a = F.linear(a, g(p + "attn.proj.weight"), g(p + "attn.proj.bias"))
        if tw.layerscale:
            a = a * g(p + "ls1.scale_factor")
        t = r(t + a, rv)     # (the default path keeps the ViT token rows in bf16)
        h = r(F.layer_norm(t, (D,), g(p + "norm2.weight"), g(p + "norm2.bias"), eps=tw.ln_eps), rv)
        f = r(F.gelu(F.linear(h, g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias"))), rv)
        f = F.linear(f, g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias"))
        if tw.layerscale:
            f = f * g(p + "ls2.scale_factor")
        t = r(t + f, rv)
    return t[:, tw.n_prefix:, :]


def vision(pix, sd, cfg, rv):
    feats = r(torch.cat([vit_tower(pix[:, 3 * i:3 * i + 3], sd, orc.TOWER_PREFIXES[i], tw, rv) for i, tw in enumerate(cfg.towers)], dim=2), rv)
    g = lambda k: sd["projector." + k]
    x = r(F.gelu(F.linear(feats, g("fc1.weight"), g("fc1.bias"))), rv)
    x = r(F.gelu(F.linear(x, g("fc2.weight"), g("fc2.bias"))), rv)
    return r(F.linear(x, g("fc3.weight"), g("fc3.bias")), rv)


def llama_layer(h, sd, li, lc, positions, kv, on, prefill):
    p = f"language_model.model.layers.{li}."
    g = lambda k: sd[p + k]
    B, T, _ = h.shape
    Hq, Hkv, hd = lc.num_heads, lc.num_kv_heads, lc.head_dim
    x = r(orc.rms_norm(h, g("input_layernorm.weight"), lc.rms_eps), "xn" in on)
    q = F.linear(x, g("self_attn.q_proj.weight")).view(B, T, Hq, hd).transpose(1, 2)
    k = F.linear(x, g("self_attn.k_proj.weight")).view(B, T, Hkv, hd).transpose(1, 2)
    v = F.linear(x, g("self_attn.v_proj.weight")).view(B, T, Hkv, hd).transpose(1, 2)
    cos, sin = orc.rope_cos_sin(positions, hd, lc.rope_theta, torch.float32)
    cos, sin = cos[None, None], sin[None, None]
    q = r(q * cos + orc._rotate_half(q) * sin, "q" in on)
    k = r(k * cos + orc._rotate_half(k) * sin, "kv" in on)
    v = r(v, "kv" in on)
    if kv is not None:
        k = torch.cat([kv[0], k], dim=2)
        v = torch.cat([kv[1], v], dim=2)
    L = k.shape[2]
    rep = Hq // Hkv
    kk = k.repeat_interleave(rep, dim=1) if rep > 1 else k
    vv = v.repeat_interleave(rep, dim=1) if rep > 1 else v
    att = torch.matmul(q, kk.transpose(2, 3)) * (hd ** -0.5)
    key_pos = torch.arange(L, device=positions.device)
    att = att.masked_fill((key_pos[None, :] > positions[:, None])[None, None], float("-inf"))
    att = r(F.softmax(att, dim=-1, dtype=torch.float32), "p" in on and prefill)
    a = r(torch.matmul(att, vv).transpose(1, 2).reshape(B, T, Hq * hd), "attn" in on)
    h = r(h + F.linear(a, g("self_attn.o_proj.weight")), "resid" in on)
    x = r(orc.rms_norm(h, g("post_attention_layernorm.weight"), lc.rms_eps), "xn" in on)
    m = r(F.silu(F.linear(x, g("mlp.gate_proj.weight"))) * F.linear(x, g("mlp.up_proj.weight")), "act" in on)
    return r(h + F.linear(m, g("mlp.down_proj.weight")), "resid" in on), (k, v)


def llama(emb, sd, lc, cache, on, prefill):
    h = emb
    T = h.shape[1]
    past = 0 if cache is None else cache[0][0].shape[2]
    pos = torch.arange(past, past + T, device=h.device)
    new = []
    for li in range(lc.num_layers):
        h, kv = llama_layer(h, sd, li, lc, pos, None if cache is None else cache[li], on, prefill)
        new.append(kv)
    h = r(orc.rms_norm(h[:, -1:], sd["language_model.model.norm.weight"], lc.rms_eps), "xn" in on)
    return F.linear(h, sd["language_model.lm_head.weight"])[0, -1], new


def trace(cfg, sd, pix, row, gen, T, on, dev):
    with torch.inference_mode():
        proj = vision(pix, sd, cfg, "vit" in on)
        emb = orc.splice(torch.tensor([row], device=dev), proj, sd)
        logits, cache = llama(emb, sd, cfg.llm, None, on, True)
        out = []
        for t in range(T):
            out.append(logits.float().cpu())
            if gen is None:
                nxt = int(logits.argmax())
            else:
                nxt = gen[t]
            if t + 1 < T:
                logits, cache = llama(orc.embed_tokens(torch.tensor([[nxt]], device=dev), sd), sd, cfg.llm, cache, on, False)
    return out


def main():
    from emmax.config import EmmaXConfig
    from emmax.weights import synthetic_state_dict

    dev = "cuda:0" if torch.cuda.is_available() else "cpu"
    T = int(os.environ.get("ABL_STEPS", "64"))
    cfg = EmmaXConfig.tiny() if os.environ.get("ABL_TINY") else EmmaXConfig.emma_x_7b()   # (ABL_TINY=1: a plumbing run of this script on the host)
    sd = synthetic_state_dict(cfg, seed=33)
    sd = {k: v.to(BF).float().to(dev) for k, v in sd.items()}
    rng = np.random.default_rng(77)
    frames = rng.integers(0, 256, size=(1, 224, 224, 3), dtype=np.uint8)
    row = [1] + [int(x) for x in rng.integers(3, 31744, size=31 if os.environ.get("ABL_TINY") else 511)]
    pix = orc.preprocess_frames(frames, cfg).to(dev)
    ref = trace(cfg, sd, pix, row, None, T, frozenset(), dev)
    gen = [int(x.argmax()) for x in ref]
    rows = {}

    def measure(name, on):
        got = trace(cfg, sd, pix, row, gen, T, frozenset(on), dev)
        e = np.array([((g - f).abs().max() / f.abs().max()).item() for g, f in zip(got, ref)])
        flips = int(sum(int(g.argmax()) != t for g, t in zip(got, gen)))
        rows[name] = {"rounded": sorted(on), "rel_err_median": float(np.median(e)), "rel_err_max": float(e.max()), "argmax_flips": flips}
        print(name, rows[name], flush=True)

    measure("none", [])
    for pt in POINTS:
        measure("only_" + pt, [pt])
    measure("all", list(POINTS))
    measure("all_but_resid", [p for p in POINTS if p != "resid"])     # = the default path since round 5 (fp32 residual stream)
    for pt in ("kv", "vit", "xn", "attn", "act"):
        measure("default_but_" + pt, [p for p in POINTS if p not in ("resid", pt)])
    out = {"what": "fp32 restatement on the GPU, Emma-X-7B shape, random weights seed 33, one frame + 512-token prompt, %d teacher-forced steps; ONE "
                   "activation class rounded to bf16 per row (tools/exact_ablation.py); errors relative to max|logit| of the step" % T,
           "points": {"xn": "RMSNorm output into qkv / gate-up / lm-head", "q": "rotated queries", "kv": "K / V rows", "p": "softmax probabilities (prefill)",
                      "attn": "attention output into o-proj", "act": "SwiGLU product into down", "resid": "residual stream", "vit": "everything inside the towers + projector"},
           "rows": rows}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_exact_ablation%s.json" % ("_tiny" if os.environ.get("ABL_TINY") else "")), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
