"""Fixture for the HuBERT-base structure (GroupNorm over time after the first conv only, no conv bias, post-LN layers): HF
HubertModel -- the runnable twin of fairseq's HuBERT, which the reference calls (slam_model.py:335-341) and which is not installed --
configured feat_extract_norm="group", do_stable_layer_norm=False, conv_bias=False, on an equal-length and a ragged zero-padded batch
(frame mask by fairseq's rule, as in make_golden.gen_hubert_ragged).  python oracle/make_golden_hubert_base.py ->
tests/golden/hubert_base_tiny.npz"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import slam_oracle as O  # noqa: E402
from oracle.make_golden import GOLD, pack, wsum  # noqa: E402
from oracle.make_golden_cases import HUBERT_BASE_TINY  # noqa: E402


def main():
    from transformers import HubertConfig, HubertModel
    cfg = HUBERT_BASE_TINY
    hc = HubertConfig(hidden_size=cfg["hub_dim"], num_hidden_layers=cfg["hub_layers"], num_attention_heads=cfg["hub_heads"],
                      intermediate_size=cfg["hub_ffn"], conv_dim=list(cfg["hub_conv_dim"]), conv_kernel=list(cfg["hub_conv_kernel"]),
                      conv_stride=list(cfg["hub_conv_stride"]), conv_bias=False, feat_extract_norm="group",
                      do_stable_layer_norm=False, feat_proj_layer_norm=True, num_conv_pos_embeddings=cfg["hub_pos_k"],
                      num_conv_pos_embedding_groups=cfg["hub_pos_groups"], hidden_dropout=0.0, attention_dropout=0.0,
                      activation_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0, mask_feature_prob=0.0,
                      layer_norm_eps=cfg["hub_eps"], hidden_act="gelu", feat_extract_activation="gelu")
    hc._attn_implementation = "eager"
    m = HubertModel(hc).eval()
    W = O.init_hubert_weights(cfg, seed=8)
    sd = m.state_dict()
    names = set()
    with torch.no_grad():
        for k in sd:
            if "pos_conv_embed.conv.parametrizations" in k or k == "masked_spec_embed":
                continue
            sd[k].copy_(W["encoder." + k])
            names.add("encoder." + k)
        w = W["encoder.encoder.pos_conv_embed.conv.weight"]
        sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"].copy_(w)
        sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"].copy_(w.norm(dim=(0, 1), keepdim=True))
    m.load_state_dict(sd)
    assert names | {"encoder.encoder.pos_conv_embed.conv.weight"} == set(W), sorted(set(W) ^ names)[:6]
    fx = {"weights_sha256": np.array(wsum(W))}
    wav = O.synth_audio(2, 1.0, seed=4323)          # hubert_base: normalize=False (raw waveform in)
    with torch.no_grad():
        out = m(wav).last_hidden_state
        mine = O.hubert_encoder(W, cfg, wav)
    print("equal-length: max |oracle - HF|", float((out - mine).abs().max()), tuple(out.shape))
    fx["wav"], fx["out_shape"] = wav.numpy(), np.array(out.shape)
    pack(fx, "out", out.numpy(), limit=65536)
    n_valid = torch.tensor([16000, 9000, 12345])
    N = 16000
    clips = O.synth_audio(3, 1.0, seed=4324)
    rag = torch.zeros(3, N)
    for b, n in enumerate(n_valid.tolist()):
        rag[b, :n] = clips[b, :n]
    amask = (torch.arange(N)[None, :] < n_valid[:, None]).long()

    def fairseq_frame_mask(feature_vector_length, attention_mask):
        return ~O.hubert_frame_padding_mask(attention_mask.shape[1], feature_vector_length, attention_mask.sum(-1))
    m._get_feature_vector_attention_mask = fairseq_frame_mask
    with torch.no_grad():
        out_r = m(rag, attention_mask=amask).last_hidden_state
        mine_r = O.hubert_encoder(W, cfg, rag, n_valid=n_valid)
    pad = O.hubert_frame_padding_mask(N, out_r.shape[1], n_valid)
    print("ragged: max |oracle - HF| on valid frames", float((out_r - mine_r).masked_fill(pad[:, :, None], 0.0).abs().max()),
          "valid frames", (~pad).sum(1).tolist())
    fx["ragged.wav"], fx["ragged.n_valid"], fx["ragged.frame_padding_mask"] = rag.numpy(), n_valid.numpy(), pad.numpy()
    pack(fx, "ragged.out", out_r.masked_fill(pad[:, :, None], 0.0).numpy(), limit=65536)
    np.savez_compressed(os.path.join(GOLD, "hubert_base_tiny.npz"), **fx)
    print("hubert_base_tiny.npz written")


if __name__ == "__main__":
    main()
