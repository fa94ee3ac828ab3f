"""Fixture for the WavLM encoder: the reference's OWN module (src/slam_llm/models/wavlm/WavLM.py + modules.py, imported UNMODIFIED
from /root/reference -- it is vendored there, unlike fairseq's HuBERT) through the reference's wrapper call
`WavLMEncoder.extract_features(source, padding_mask)` (models/encoder.py:126-127), eval mode, on an equal-length and on a ragged
zero-padded batch; the oracle restatement must agree.  Run in the build container: python oracle/make_golden_wavlm.py ->
tests/golden/wavlm_tiny.npz (WavLM-Large structure) and tests/golden/wavlm_base_tiny.npz (Base: group-norm extractor, post-LN)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import slam_oracle as O  # noqa: E402
from oracle.make_golden import GOLD, pack, wsum  # noqa: E402  (also puts /root/reference/src on sys.path with its stubs)
from oracle.make_golden_cases import WAVLM_BASE_TINY, WAVLM_TINY  # noqa: E402


def one_case(c, name, seed, wav_norm):
    from slam_llm.models.encoder import WavLMEncoder
    from slam_llm.models.wavlm.WavLM import WavLM, WavLMConfig
    layers = "[" + ", ".join(f"({co},{k},{s})" for co, k, s in zip(c["hub_conv_dim"], c["hub_conv_kernel"], c["hub_conv_stride"])) + "]"
    cfg = WavLMConfig(dict(extractor_mode=c["hub_extractor_mode"], encoder_layers=c["hub_layers"], encoder_embed_dim=c["hub_dim"],
                           encoder_ffn_embed_dim=c["hub_ffn"], encoder_attention_heads=c["hub_heads"],
                           layer_norm_first=c["hub_layer_norm_first"],
                           conv_feature_layers=layers, conv_bias=False, normalize=wav_norm, conv_pos=c["hub_pos_k"],
                           conv_pos_groups=c["hub_pos_groups"], relative_position_embedding=True, num_buckets=c["wavlm_buckets"],
                           max_distance=c["wavlm_max_distance"], gru_rel_pos=True))
    model = WavLM(cfg)
    W = O.init_wavlm_weights(c, seed=seed)
    sd = {k[len("encoder.model."):]: v for k, v in W.items()}
    missing, unexpected = model.load_state_dict(sd, strict=True)
    enc = WavLMEncoder(cfg, model).eval()
    assert set("encoder.model." + k for k in model.state_dict()) == set(W), "oracle weight names != the reference module's"
    fx = {"weights_sha256": np.array(wsum(W))}
    norm = (lambda t, n: torch.nn.functional.layer_norm(t, (n,))) if wav_norm else (lambda t, n: t)
    wav = norm(O.synth_audio(3, 1.0, seed=21), 16000)
    with torch.no_grad():
        out = enc.extract_features(wav, torch.zeros(wav.shape, dtype=torch.bool))
        mine = O.wavlm_encoder(W, c, wav)
    print(name, "equal-length: max |oracle - reference|", float((out - mine).abs().max()), "out", tuple(out.shape))
    fx["wav"] = wav.numpy()
    fx["out_shape"] = np.array(out.shape)
    pack(fx, "out", out.numpy(), limit=65536)
    # ragged: clips of 16000 / 9000 / 12345 samples, zero padded (the dataset normalises each clip BEFORE padding)
    nv = torch.tensor([16000, 9000, 12345])
    rag = torch.zeros(3, 16000)
    raw = O.synth_audio(3, 1.0, seed=22)
    for b_, n in enumerate(nv.tolist()):
        rag[b_, :n] = norm(raw[b_, :n], n)
    pm = torch.arange(16000)[None, :] >= nv[:, None]
    with torch.no_grad():
        out_r = enc.extract_features(rag, pm)
        mine_r = O.wavlm_encoder(W, c, rag, n_valid=nv)
    fpad = O.hubert_frame_padding_mask(16000, out_r.shape[1], nv)
    d_ = (out_r - mine_r).masked_fill(fpad[:, :, None], 0.0)
    print(name, "ragged: max |oracle - reference| on valid frames", float(d_.abs().max()), "valid frames", (~fpad).sum(1).tolist())
    fx["ragged.wav"], fx["ragged.n_valid"] = rag.numpy(), nv.numpy()
    fx["ragged.frame_padding_mask"] = fpad.numpy()
    pack(fx, "ragged.out", out_r.masked_fill(fpad[:, :, None], 0.0).numpy(), limit=65536)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **fx)
    print(name + ".npz written")


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "large"):
        one_case(WAVLM_TINY, "wavlm_tiny", 9, True)
    if which in ("all", "base"):     # WavLM Base: the released cfg has normalize=False (raw waveform in)
        one_case(WAVLM_BASE_TINY, "wavlm_base_tiny", 10, False)


if __name__ == "__main__":
    main()
