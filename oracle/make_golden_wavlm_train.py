"""Fixture for the UN-FROZEN WavLM encoder's train-mode regularisers -- TEST INFRASTRUCTURE.  The reference's own module
(src/slam_llm/models/wavlm/WavLM.py + modules.py, imported UNMODIFIED from /root/reference) in .train() mode through the reference's
wrapper call `WavLMEncoder.extract_features(source, padding_mask)` (models/encoder.py:126-127; models/slam_model.py:317-318 leaves the
encoder in train mode when train_config.freeze_encoder is false): dropout_input, the dropout after the positional conv, per layer
attention_dropout / dropout1 / dropout2 / dropout3 and encoder_layerdrop (WavLM.py:180-185, 353, 584, 596-597, 702-726).

torch's dropout RNG cannot be shared with a device kernel, so the masks are made an INPUT: torch.nn.functional.dropout is replaced by a
recorder that draws its own Bernoulli mask, applies it and keeps it, and torch.nn.functional.scaled_dot_product_attention (which
F.multi_head_attention_forward -- the call at modules.py:540-565 -- uses when need_weights is false, and which applies attention dropout
internally) by its definition softmax(q k^T / sqrt(d) + mask) -> dropout -> @ v, so that the mask on the probabilities is recorded too.
Nothing of the reference is modified.  Layerdrop reads numpy's global stream (np.random.random(), one draw per layer): the seeds below
are chosen so that case A skips layer 1 and case B skips layer 0 (then position_bias is never created and the remaining layers run
without the bias, WavLM.py:593-599).

Run in the build container: python oracle/make_golden_wavlm_train.py -> tests/golden/wavlm_train_tiny.npz; the oracle restatement with
the same masks must reproduce output and every parameter gradient (checked here and in tests/test_oracle_golden.py)."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import slam_oracle as O  # noqa: E402
from oracle.make_golden import GOLD, pack, wsum  # noqa: E402  (also puts /root/reference/src on sys.path with its stubs)
from oracle.make_golden_cases import WAVLM_TRAIN_REG, WAVLM_TRAIN_TINY  # noqa: E402

import torch.nn.functional as F  # noqa: E402


class Recorder:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.masks = []

    def dropout(self, input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        m = (torch.rand(input.shape, generator=self.g) >= p).to(input.dtype) / (1.0 - p)
        self.masks.append(m)
        return input * m

    def sdpa(self, q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **kw):
        assert not is_causal
        sc = (q @ k.transpose(-2, -1)) * (scale if scale is not None else 1.0 / math.sqrt(q.shape[-1]))
        if attn_mask is not None:
            sc = sc.masked_fill(~attn_mask, float("-inf")) if attn_mask.dtype == torch.bool else sc + attn_mask
        pr = torch.softmax(sc, dim=-1)
        if dropout_p > 0.0:
            pr = self.dropout(pr, dropout_p, True)
        return pr @ v


def find_seed(pattern, layerdrop):
    """numpy seed whose first len(pattern) draws keep (u > layerdrop) / skip exactly as `pattern` says"""
    for s in range(1000):
        np.random.seed(s)
        if tuple(bool(np.random.random() > layerdrop) for _ in pattern) == tuple(pattern):
            return s
    raise RuntimeError("no seed")


def build(c, seed=9):
    from slam_llm.models.encoder import WavLMEncoder
    from slam_llm.models.wavlm.WavLM import WavLM, WavLMConfig
    layers = "[" + ", ".join(f"({co},{k},{s})" for co, k, s in zip(c["hub_conv_dim"], c["hub_conv_kernel"], c["hub_conv_stride"])) + "]"
    cfg = WavLMConfig(dict(extractor_mode=c["hub_extractor_mode"], encoder_layers=c["hub_layers"], encoder_embed_dim=c["hub_dim"],
                           encoder_ffn_embed_dim=c["hub_ffn"], encoder_attention_heads=c["hub_heads"], layer_norm_first=c["hub_layer_norm_first"],
                           conv_feature_layers=layers, conv_bias=False, normalize=c["hub_layer_norm_first"], conv_pos=c["hub_pos_k"],
                           conv_pos_groups=c["hub_pos_groups"], relative_position_embedding=True, num_buckets=c["wavlm_buckets"],
                           max_distance=c["wavlm_max_distance"], gru_rel_pos=True, **WAVLM_TRAIN_REG))
    model = WavLM(cfg)
    W = O.init_wavlm_weights(c, seed=seed)
    model.load_state_dict({k[len("encoder.model."):]: v for k, v in W.items()}, strict=True)
    return WavLMEncoder(cfg, model), W


def train_dict(masks, kept, n_layers):
    """recorded masks, in call order, -> the oracle's `train` argument"""
    it = iter(masks)
    tr = {"input": next(it), "x": next(it), "layers": []}
    for i in range(n_layers):
        tr["layers"].append(dict(attn=next(it), d1=next(it), d2=next(it), d3=next(it)) if kept[i] else None)
    assert next(it, None) is None
    return tr


def one_case(fx, tag, pattern, ragged, c=WAVLM_TRAIN_TINY, seed=9):
    enc, W = build(c, seed)
    enc.train()
    N = 16000
    wav = O.synth_audio(2, 1.0, seed=31)
    if c["hub_layer_norm_first"]:       # the large checkpoints' cfg has normalize=True (dataset-side layer norm), the base ones have not
        wav = torch.nn.functional.layer_norm(wav, (N,))
    nv = torch.tensor([N, 11200] if ragged else [N, N])
    if ragged:
        wav[1, int(nv[1]):] = 0.0
    pm = torch.arange(N)[None, :] >= nv[:, None]
    seed = find_seed(pattern, WAVLM_TRAIN_REG["encoder_layerdrop"])
    rec = Recorder(1000 + seed)
    ran = []
    hooks = [l.register_forward_hook(lambda m, i, o, k=k: ran.append(k)) for k, l in enumerate(enc.model.encoder.layers)]
    orig = (F.dropout, F.scaled_dot_product_attention)
    F.dropout, F.scaled_dot_product_attention = rec.dropout, rec.sdpa
    try:
        np.random.seed(seed)
        out = enc.extract_features(wav, pm)
    finally:
        F.dropout, F.scaled_dot_product_attention = orig
        for h in hooks:
            h.remove()
    kept = [k in ran for k in range(c["hub_layers"])]
    assert tuple(kept) == tuple(pattern), (kept, pattern)
    assert len(rec.masks) == 2 + 4 * sum(kept), len(rec.masks)
    B, T, d = out.shape
    H = c["hub_heads"]
    masks = [m.reshape(B, H, T, T) if m.numel() == B * H * T * T and m.dim() != 3 else m for m in rec.masks]
    tr = train_dict(masks, kept, c["hub_layers"])
    for lm in tr["layers"]:
        if lm is not None:
            lm["attn"] = lm["attn"].reshape(B, H, T, T)
            lm["d1"], lm["d3"] = lm["d1"].transpose(0, 1), lm["d3"].transpose(0, 1)      # the layers run T x B x C
            lm["d2"] = lm["d2"].transpose(0, 1)
    cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)) * 0.1
    fpad = O.hubert_frame_padding_mask(N, T, nv)
    cot = cot.masked_fill(fpad[:, :, None], 0.0)       # padded frames' rows are never read downstream
    (out * cot).sum().backward()
    ref_grads = {"encoder.model." + k: (p.grad.clone() if p.grad is not None else None) for k, p in enc.model.named_parameters()}
    # the oracle with the same masks
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    mine = O.wavlm_encoder(Wg, c, wav, n_valid=nv if ragged else None, train=tr)
    (mine * cot).sum().backward()
    d_ = (out - mine).masked_fill(fpad[:, :, None], 0.0)
    worst = 0.0
    gmax = max(float(g.abs().max()) for g in ref_grads.values() if g is not None)
    for k, g in ref_grads.items():
        mg = Wg[k].grad
        if g is None or float(g.abs().max()) == 0.0:
            assert mg is None or float(mg.abs().max()) == 0.0, k
            continue
        if k.endswith("k_proj.bias") and float(g.abs().max()) < 1e-5 * gmax:     # key biases: mathematically zero gradient (softmax shift)
            assert float(mg.abs().max()) < 1e-5 * gmax, k
            continue
        r_ = float((g - mg).abs().max() / (g.abs().max() + 1e-12))
        if r_ > 1e-3:
            print("   ", k, r_, float(g.abs().max()), float(mg.abs().max()))
        worst = max(worst, r_)
    print(f"{tag}: kept {kept}, {len(rec.masks)} masks, max |oracle - reference| out {float(d_.detach().abs().max()):.2e}, worst relative grad diff {worst:.2e}")
    assert float(d_.abs().max()) < 1e-4 and worst < 1e-3
    P = tag + "."
    fx[P + "wav"], fx[P + "n_valid"], fx[P + "kept"], fx[P + "np_seed"] = wav.numpy(), nv.numpy(), np.array(kept), np.int64(seed)
    fx[P + "cot"] = cot.numpy()
    fx[P + "out_shape"] = np.array(out.shape)
    pack(fx, P + "out", out.detach().masked_fill(fpad[:, :, None], 0.0).numpy(), limit=65536)
    names = ["input", "x"] + [f"l{i}.{k}" for i in range(c["hub_layers"]) if kept[i] for k in ("attn", "d1", "d2", "d3")]
    flat = [tr["input"], tr["x"]] + [tr["layers"][i][k] for i in range(c["hub_layers"]) if kept[i] for k in ("attn", "d1", "d2", "d3")]
    for n, m in zip(names, flat):
        fx[P + "mask." + n + ".shape"] = np.array(m.shape)
        fx[P + "mask." + n] = np.packbits((m != 0).numpy().reshape(-1))
    for k, g in ref_grads.items():
        if g is None:
            fx[P + "grad." + k + ".__none"] = np.int64(1)
        else:
            pack(fx, P + "grad." + k, g.numpy(), limit=2048)


def main():
    fx = {"weights_sha256": np.array(wsum(O.init_wavlm_weights(WAVLM_TRAIN_TINY, seed=9)))}
    one_case(fx, "A", (True, False, True), ragged=True)        # layer 1 skipped, ragged batch
    one_case(fx, "B", (False, True, True), ragged=False)       # layer 0 skipped: no position bias at all
    one_case(fx, "C", (True, True, True), ragged=False)
    # Base / Base+ structure (group-norm extractor, post-LN layers: the dropout after the positional conv sits behind the encoder LayerNorm,
    # WavLM.py:582-584; dropout1 / dropout3 before the residual adds that the LayerNorms follow, :716-739)
    from oracle.make_golden_cases import WAVLM_BASE_TINY
    fx["base.weights_sha256"] = np.array(wsum(O.init_wavlm_weights(WAVLM_BASE_TINY, seed=10)))
    one_case(fx, "D", (True, True), ragged=True, c=WAVLM_BASE_TINY, seed=10)
    one_case(fx, "E", (True, False), ragged=False, c=WAVLM_BASE_TINY, seed=10)
    np.savez_compressed(os.path.join(GOLD, "wavlm_train_tiny.npz"), **fx)
    print("wavlm_train_tiny.npz written", os.path.getsize(os.path.join(GOLD, "wavlm_train_tiny.npz")), "bytes")


if __name__ == "__main__":
    main()
