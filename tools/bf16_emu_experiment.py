"""CPU experiment (no GPU): which bf16 rounding site carries the q_proj-adapter degradation at depth?

fp32 oracle vs the bf16-emulating twin (oracle/bf16_emulation.py) on a deep random-init Llama at reduced or true width:
per-tensor gradient cosine of (emulated vs fp32), (emulated vs emulated with re-ordered sums), and single-site ablations.
    python tools/bf16_emu_experiment.py --dim 1024 --layers 32 [--ablate]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bf16_emulation as E  # noqa: E402
from oracle import slam_oracle as O  # noqa: E402


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


def grads_of(W, names, fn):
    for n in names:
        W[n].requires_grad_(True)
    loss, _ = fn()
    loss.backward()
    g = {n: W[n].grad.detach().clone() for n in names}
    for n in names:
        W[n].requires_grad_(False)
        W[n].grad = None
    return float(loss.detach()), g


def summarize(tag, ga, gb, names, out):
    rows = []
    for n in names:
        rows.append((n, 1.0 - cosine(ga[n], gb[n])))
    qa = [d for n, d in rows if "q_proj.lora_A" in n]
    qb = [d for n, d in rows if "q_proj.lora_B" in n]
    va = [d for n, d in rows if "v_proj.lora_A" in n]
    vb = [d for n, d in rows if "v_proj.lora_B" in n]
    worst = max(rows, key=lambda r: r[1])
    rec = dict(tag=tag, worst=worst[1], worst_name=worst[0], qA_max=max(qa), qA_mean=float(np.mean(qa)), qB_max=max(qb),
               vA_max=max(va), vB_max=max(vb))
    out.append(rec)
    print(json.dumps(rec), flush=True)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--T", type=int, default=380)
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--vocab", type=int, default=8192)
    ap.add_argument("--ablate", action="store_true")
    ap.add_argument("--only", default="", help="comma list of single sites to switch ON alone")
    ap.add_argument("--without", default="", help="comma list of single sites to switch OFF alone")
    ap.add_argument("--preround-weights", action="store_true", help="round the frozen weights to bf16 IN PLACE first (both runs use them; saves the rounded copies)")
    ap.add_argument("--heads", type=int, default=0)
    ap.add_argument("--kv-heads", type=int, default=0)
    ap.add_argument("--ffn", type=int, default=0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    d = a.dim
    cfg = O.make_config(llm_dim=d, llm_layers=a.layers, llm_heads=a.heads or d // 128, llm_kv_heads=a.kv_heads or max(1, d // 512), llm_head_dim=128,
                        llm_ffn=a.ffn or d * 7 // 2, vocab=a.vocab, rope_theta=500000.0, lora_r=16, lora_alpha=32, enc_layers=0)
    W = O.init_weights(cfg, seed=42)
    ALL = E.ALL_SITES
    if a.preround_weights:
        for n, t in W.items():
            if "lora_" not in n and t.dim() == 2:
                t.copy_(E.rb(t))
        ALL = E.ALL_SITES - {"weights"}
    g = torch.Generator().manual_seed(7)
    # inputs shaped like the C3 step: 300 "audio" rows from a projector-like distribution, then token embeddings
    emb = W["llm.base_model.model.model.embed_tokens.weight"]
    ids = torch.randint(3, a.vocab, (a.B, a.T), generator=g)
    embeds = emb[ids].clone()
    embeds[:, :300] = torch.randn(a.B, 300, d, generator=g) * 0.5
    am = torch.ones(a.B, a.T, dtype=torch.long)
    labels = torch.full((a.B, a.T), -100, dtype=torch.long)
    labels[:, 316:] = ids[:, 316:]
    names = [n for n in W if "lora_" in n]
    out = []
    t0 = time.time()
    l32, g32 = grads_of(W, names, lambda: O.llama_forward(W, cfg, embeds, am, labels))
    print(f"fp32 loss {l32:.5f} ({time.time() - t0:.1f} s)", flush=True)
    lem, gem = grads_of(W, names, lambda: E.llama_forward_emulated(W, cfg, embeds, am, labels, sites=ALL))
    print(f"emulated loss {lem:.5f}", flush=True)
    rows = summarize("emulated_vs_fp32", gem, g32, names, out)
    lre, gre = grads_of(W, names, lambda: E.llama_forward_emulated(W, cfg, embeds, am, labels, sites=ALL, reorder=True))
    print(f"emulated (re-ordered sums) loss {lre:.5f}", flush=True)
    summarize("reordered_vs_fp32", gre, g32, names, out)
    summarize("emulated_vs_reordered", gem, gre, names, out)
    if a.ablate:
        for s in sorted(ALL):
            _, gs = grads_of(W, names, lambda: E.llama_forward_emulated(W, cfg, embeds, am, labels, sites=ALL - {s}))
            summarize(f"without_{s}_vs_fp32", gs, g32, names, out)
        for s in sorted(ALL):
            _, gs = grads_of(W, names, lambda: E.llama_forward_emulated(W, cfg, embeds, am, labels, sites={s}))
            summarize(f"only_{s}_vs_fp32", gs, g32, names, out)
    for s in [x for x in a.without.split(",") if x]:
        _, gs = grads_of(W, names, lambda: E.llama_forward_emulated(W, cfg, embeds, am, labels, sites=ALL - {s}))
        summarize(f"without_{s}_vs_fp32", gs, g32, names, out)
    for s in [x for x in a.only.split(",") if x]:
        _, gs = grads_of(W, names, lambda: E.llama_forward_emulated(W, cfg, embeds, am, labels, sites={s}))
        summarize(f"only_{s}_vs_fp32", gs, g32, names, out)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(dict(args=vars(a), results=out, per_tensor_emulated_vs_fp32=rows), f, indent=1)


if __name__ == "__main__":
    main()
