"""GPU probe: gradients of WavLM's gate parameters (grep_linear / grep_a) and bucket table, HIP encoder vs oracle, with and without the
train-mode regularisers and at two scales of the bucket embedding."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import slam_oracle as O  # noqa: E402
from oracle.make_golden_cases import WAVLM_TRAIN_TINY as C  # noqa: E402
from tests import golden_util as G  # noqa: E402
from slam_llm_amd import ops  # noqa: E402
from slam_llm_amd.model import HipWavLMEncoder, TrainableStore  # noqa: E402

dev = torch.device("cuda:0")
fx = G.load("wavlm_train_tiny")
RB = "encoder.model.encoder.layers.0.self_attn.relative_attention_bias.weight"


def run(p, p_attn, scale, seed):
    torch.manual_seed(seed)
    reg = dict(hub_dropout=p, hub_attention_dropout=p_attn, hub_activation_dropout=p, hub_dropout_input=p, hub_layerdrop=0.0)
    W = O.init_wavlm_weights(C, seed=9)
    W[RB] = W[RB] * scale
    store = TrainableStore(dev)
    enc = HipWavLMEncoder(dict(C, **reg), dev, store=store)
    store.allocate(); enc.bind(); enc.load(W); store.refresh_bf16(); enc.refresh(); enc.train()
    wav = torch.from_numpy(fx["C.wav"])
    stash = {}
    out = enc.forward_train(wav.to(dev), stash, None)
    S = stash["encoder"]
    B, T, d, H = out.shape[0], out.shape[1], C["hub_dim"], C["hub_heads"]
    Tp = (T + 63) // 64 * 64
    ones = lambda n: torch.ones((B * T, n), dtype=torch.bfloat16, device=dev)      # noqa: E731
    hid = lambda key, n=d: None if key is None else (ops.dropout(ones(n), *key).float().cpu().view(B, T, n).ne(0).float() / (1 - key[0]))      # noqa: E731
    tr = {"input": hid(S["k_in"]), "x": hid(S["k_x"]), "layers": []}
    for R in S["blocks"]:
        am = None if R["ka"] is None else torch.from_numpy(G.attn_keep_mask(R["ka"][1], R["ka"][0], B, H, T, T, Tp, Tp)) / (1 - R["ka"][0])
        tr["layers"].append(dict(attn=am, d1=hid(R["k1"]), d2=hid(R["k2"], C["hub_ffn"]), d3=hid(R["k3"])))
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    ref = O.wavlm_encoder(Wg, C, wav, train=tr)
    cot = torch.from_numpy(fx["C.cot"])
    (ref * cot).sum().backward()
    enc.backward_hip(cot.to(dev).to(torch.bfloat16).reshape(B * T, d).contiguous(), stash, acc=False)
    print(f"p={p} p_attn={p_attn} table x{scale} seed {seed}: out cosine {G.cosine(ref.detach().numpy(), out.float().cpu().numpy()):.5f}")
    for n in W:
        if "grep_" in n or "relative_attention" in n or n.endswith("layers.0.self_attn.q_proj.weight"):
            gold, mine = Wg[n].grad, store.grad_view(n).float().cpu().reshape(W[n].shape)
            print(f"   {n[len('encoder.model.encoder.'):]:55s} cos {G.cosine(gold.numpy(), mine.numpy()):8.5f}  |g| {float(gold.norm()):.3e} mine {float(mine.norm()):.3e}")


for (p, pa, sc) in ((0.0, 0.0, 1.0), (0.0, 0.0, 25.0), (0.1, 0.0, 25.0), (0.0, 0.1, 25.0)):
    run(p, pa, sc, 1)
