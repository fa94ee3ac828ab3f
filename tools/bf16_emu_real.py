"""CPU experiment (no GPU): the full-depth C3 parity case of tests/test_headline_gpu.py (Whisper-large-v3 x E layers -> projector ->
Llama-3-8B x L layers, B clips x 30 s, T = 380, seeds of the test) through the fp32 oracle and through the bf16-emulating twin
(oracle/bf16_emulation.py): per-tensor gradient deviation 1 - cos of emulated-vs-fp32, of two emulations that differ only in summation
order, and with single rounding sites switched off / on.
    python tools/bf16_emu_real.py --enc-layers 32 --llm-layers 32 --B 2 --without h,x,ds --only h --out profiles/r06_bf16_emulation_cpu.json
Needs ~50 GB of host RAM at full depth (the frozen weights are rounded IN PLACE after the fp32 run)."""
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
    loss, logits = fn()
    loss.backward()
    g = {n: W[n].grad.detach().clone() for n in names}
    for n in names:
        W[n].requires_grad_(False)
        W[n].grad = None
    return float(loss.detach()), g


def summarize(tag, ga, gb, names, out):
    rows = [(n, 1.0 - cosine(ga[n], gb[n])) for n in names]
    pick = lambda key: [d for n, d in rows if key in n]  # noqa: E731
    worst = max(rows, key=lambda r: r[1])
    rec = dict(tag=tag, worst=worst[1], worst_name=worst[0], qA_max=max(pick("q_proj.lora_A")), qA_mean=float(np.mean(pick("q_proj.lora_A"))),
               qB_max=max(pick("q_proj.lora_B")), vA_max=max(pick("v_proj.lora_A")), vB_max=max(pick("v_proj.lora_B")),
               proj_max=max(pick("encoder_projector")))
    out.append(rec)
    print(json.dumps(rec), flush=True)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--enc-layers", type=int, default=32)
    ap.add_argument("--llm-layers", type=int, default=32)
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--only", default="")
    ap.add_argument("--without", default="")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    cfg = O.make_config(n_mels=128, enc_dim=1280, enc_heads=20, enc_layers=a.enc_layers, llm_dim=4096, llm_layers=a.llm_layers, llm_heads=32,
                        llm_kv_heads=8, llm_head_dim=128, llm_ffn=14336, vocab=128256, rope_theta=500000.0, rms_eps=1e-5, lora_r=16, lora_alpha=32,
                        lora_targets=("q_proj", "v_proj"), lora_dropout=0.0)
    W = O.init_weights(cfg, seed=42)
    audio = O.synth_audio(a.B, 30.0, seed=1234)
    ob = O.synth_batch(cfg, audio, prompt_len=16, answer_lens=(64,), seed=1236, left_pad=False, pad_to_30s=True)
    t0 = time.time()
    with torch.no_grad():
        enc = O.whisper_encoder(W, cfg, ob["audio_mel"].permute(0, 2, 1))
    print(f"encoder {time.time() - t0:.1f} s; frame-to-frame cosine of its output {cosine(enc[0, :-1], enc[0, 1:]):.4f}", flush=True)
    emb_w = W["llm.base_model.model.model.embed_tokens.weight"]
    names = O.trainable_names(W)

    def fp32():
        proj = O.projector_concat(W, enc, cfg["ds_rate"])
        emb = O.embed_splice(emb_w, ob["input_ids"].clone(), ob["modality_mask"].bool(), proj)
        return O.llama_forward(W, cfg, emb, ob["attention_mask"], ob["labels"])

    def emulated(sites, reorder=False):
        def f():
            proj = E.projector_concat_emulated(W, E.rb(enc), cfg["ds_rate"], sites=sites, reorder=reorder)
            emb = O.embed_splice(emb_w, ob["input_ids"].clone(), ob["modality_mask"].bool(), proj)
            return E.llama_forward_emulated(W, cfg, emb, ob["attention_mask"], ob["labels"], sites=sites, reorder=reorder)
        return f

    out = []
    t0 = time.time()
    l32, g32 = grads_of(W, names, fp32)
    print(f"fp32 loss {l32:.5f} ({time.time() - t0:.1f} s)", flush=True)
    for n, t in W.items():          # the frozen matrices become their bf16 values, in place (the 'weights' site of every run below)
        if not any(m in n for m in O.TRAINABLE_MARKERS) and t.dim() >= 2:
            t.copy_(E.rb(t))
    ALL = E.ALL_SITES - {"weights"}
    t0 = time.time()
    lw, gw = grads_of(W, names, fp32)
    print(f"fp32 arithmetic on bf16 frozen weights: loss {lw:.5f} ({time.time() - t0:.1f} s)", flush=True)
    summarize("only_weights_vs_fp32", gw, g32, names, out)
    lem, gem = grads_of(W, names, emulated(ALL))
    print(f"emulated loss {lem:.5f}", flush=True)
    rows = summarize("emulated_vs_fp32", gem, g32, names, out)
    lre, gre = grads_of(W, names, emulated(ALL, reorder=True))
    print(f"emulated (re-ordered sums) loss {lre:.5f}", flush=True)
    summarize("reordered_vs_fp32", gre, g32, names, out)
    summarize("emulated_vs_reordered", gem, gre, names, out)
    for s in [x for x in a.without.split(",") if x]:
        _, gs = grads_of(W, names, emulated(ALL - {s}))
        summarize(f"without_{s}_vs_fp32", gs, g32, names, out)
    for s in [x for x in a.only.split(",") if x]:
        _, gs = grads_of(W, names, emulated({s}))
        summarize(f"only_{s}_vs_fp32", gs, g32, names, out)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(dict(args=vars(a), losses=dict(fp32=l32, bf16_weights=lw, emulated=lem, reordered=lre), results=out,
                           per_tensor_emulated_vs_fp32=rows), f, indent=1)


if __name__ == "__main__":
    main()
