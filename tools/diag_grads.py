"""GPU diagnostic: per-tensor gradient agreement (HIP bf16 path vs CPU fp32 oracle) + intermediate activations."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import slam_oracle as O  # noqa: E402
from oracle.make_golden_cases import CASES  # noqa: E402
from slam_llm_amd import ops  # noqa: E402
from slam_llm_amd.model import SlamHipModel  # noqa: E402
from tests import golden_util as G  # noqa: E402

dev = torch.device("cuda:0")


def cos(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))


for name in (sys.argv[1:] or list(CASES)):
    cfg = CASES[name]["cfg"]
    fx = G.load(name)
    W = O.init_weights(cfg, seed=42)
    model = SlamHipModel(dict(cfg, lora_dropout=0.0), dev).load_weights(W)
    model.train()
    b = {k[len("batch."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("batch.")}
    # oracle with grads on intermediates
    Wg = {k: v.clone() for k, v in W.items()}
    for n in O.trainable_names(Wg):
        Wg[n].requires_grad_(True)
    enc = O.whisper_encoder(Wg, cfg, b["audio_mel"].permute(0, 2, 1))
    proj = O.projector_concat(Wg, enc, cfg["ds_rate"])
    proj.retain_grad()
    ids = b["input_ids"].clone()
    emb = O.embed_splice(Wg["llm.base_model.model.model.embed_tokens.weight"], ids, b["modality_mask"].bool(), proj)
    emb.retain_grad()
    loss, logits = O.llama_forward(Wg, cfg, emb, b["attention_mask"], b["labels"])
    loss.backward()
    # HIP
    cap = {}
    orig = ops.embed_splice_bwd

    def wrapped(spans, dx2d, B, T, Ta, d):
        cap["dh0"] = dx2d.clone()
        out = orig(spans, dx2d, B, T, Ta, d)
        cap["dproj"] = out.clone()
        return out

    ops.embed_splice_bwd = wrapped
    import slam_llm_amd.model as M
    gb = {k: v.to(dev) for k, v in b.items()}
    outputs, acc = model(**gb)
    outputs.loss.backward()
    ops.embed_splice_bwd = orig
    print(f"== {name}: loss hip {float(outputs.loss):.5f} oracle {float(loss):.5f}")
    am = b["attention_mask"].bool()
    dh0 = cap["dh0"].float().cpu().view(emb.shape)
    print(f"  d_embeds (valid rows) cos {cos(dh0[am], emb.grad[am]):.5f} norm ratio {float(dh0[am].norm() / emb.grad[am].norm()):.4f}")
    print(f"  d_embeds (pad rows) hip norm {float(dh0[~am].norm()):.3e} oracle {float(emb.grad[~am].norm()):.3e}")
    dp = cap["dproj"].float().cpu().view(proj.shape)
    print(f"  d_proj cos {cos(dp, proj.grad):.5f} norm ratio {float(dp.norm() / proj.grad.norm()):.4f}")
    for n, p in model.store.params.items():
        g, r = p.grad.float().cpu(), Wg[n].grad
        print(f"  {cos(g, r):.5f}  nr {float(g.norm() / (r.norm() + 1e-30)):.4f}  |g| {float(r.norm()):.3e}  {n}")
