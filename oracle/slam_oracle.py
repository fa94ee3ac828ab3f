"""CPU oracle for the SLAM-LLM hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch fp32 restatement of the reference's algorithm for the path
speech-encoder -> projector -> (frozen LLM + LoRA) training step.  It is the checker for the HIP kernels:
only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the product
(`slam_llm_amd/`) never does and has no CPU fallback.

PARITY PINNING.  The reference's own test-suite pins nothing on this path ("parity
unpinned" by the reference).  The oracle is therefore pinned against outputs of the reference ITSELF run in the
authoring container: `oracle/make_golden.py` imports the reference's `slam_model.forward`
(src/slam_llm/models/slam_model.py:283-407) and `EncoderProjectorConcat` (models/projector.py:5-27) unmodified,
drives HF `WhisperEncoder` submodules with the reference's variable-length forward (models/encoder.py:13-30),
HF `LlamaForCausalLM`, a peft-0.6.0-equivalent LoRA wrapper and `torch.optim.AdamW`, and stores the results
under `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this file against those fixtures.

Third-party arithmetic that is NOT under /root/reference and is restated here from its published algorithm:
  openai-whisper (unpinned)   audio.py log_mel_spectrogram / model.py AudioEncoder
  transformers  (v4.35.2)     LlamaForCausalLM, loss
  peft          (v0.6.0)      LoRA Linear
  torch         (2.0.1)       AdamW, LambdaLR
Weights are a flat {name: fp32 tensor} dict using the reference's state_dict names.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE, N_FFT, HOP, N_SAMPLES = 16000, 400, 160, 480000


# ---------------------------------------------------------------------------------------------- a1: log-mel
def mel_filters(n_mels: int) -> torch.Tensor:
    """librosa slaney mel filterbank [n_mels, 201] (openai-whisper assets/mel_filters.npz; HF twin:
    transformers/audio_utils.py mel_filter_bank(norm='slaney', mel_scale='slaney'))."""
    def hz2mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) * (27.0 / np.log(6.4)), 3.0 * f / 200.0)

    def mel2hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), 200.0 * m / 3.0)

    freqs = np.linspace(0, SAMPLE_RATE / 2, N_FFT // 2 + 1)
    pts = mel2hz(np.linspace(hz2mel(0.0), hz2mel(SAMPLE_RATE / 2), n_mels + 2))
    fdiff = np.diff(pts)
    ramps = pts[:, None] - freqs[None, :]
    w = np.maximum(0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    w *= (2.0 / (pts[2:] - pts[:-2]))[:, None]
    return torch.from_numpy(w.astype(np.float32))


def pad_or_trim(audio: torch.Tensor, length: int = N_SAMPLES) -> torch.Tensor:
    """whisper.pad_or_trim (called at src/slam_llm/datasets/speech_dataset.py:101)."""
    n = audio.shape[-1]
    if n > length:
        return audio[..., :length]
    if n < length:
        return F.pad(audio, (0, length - n))
    return audio


def log_mel_spectrogram(audio: torch.Tensor, n_mels: int) -> torch.Tensor:
    """whisper.log_mel_spectrogram (speech_dataset.py:103); returns [n_mels, n_frames].  The per-clip
    `max - 8` floor is taken over all frames incl. the zero-padded tail."""
    window = torch.hann_window(N_FFT)
    stft = torch.stft(audio.float(), N_FFT, HOP, window=window, return_complex=True)
    mag = stft[..., :-1].abs() ** 2
    mel = mel_filters(n_mels) @ mag
    log_spec = torch.clamp(mel, min=1e-10).log10()
    if log_spec.dim() == 2:
        log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    else:
        log_spec = torch.maximum(log_spec, log_spec.amax(dim=(-2, -1), keepdim=True) - 8.0)
    return (log_spec + 4.0) / 4.0


# ---------------------------------------------------------------------------------------------- a2: whisper encoder
def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """openai-whisper model.py sinusoids(): the encoder's fixed positional embedding."""
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([t.sin(), t.cos()], dim=1)


def _ln(x, w, b, eps=1e-5):
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, eps).type(x.dtype)


def whisper_encoder(W: Dict[str, torch.Tensor], cfg: dict, mel_bct: torch.Tensor, prefix="encoder.") -> torch.Tensor:
    """extract_variable_length_features, src/slam_llm/models/encoder.py:13-30, on openai-whisper's
    AudioEncoder (pre-LN blocks; key projection has no bias; q,k each scaled by hd^-0.25; no mask).
    mel_bct: [B, n_mels, T] -> [B, ceil(T/2), d]."""
    H = cfg["enc_heads"]
    x = F.gelu(F.conv1d(mel_bct, W[prefix + "conv1.weight"], W[prefix + "conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, W[prefix + "conv2.weight"], W[prefix + "conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1)
    x = (x + W[prefix + "positional_embedding"][: x.shape[1]]).to(x.dtype)
    B, T, d = x.shape
    hd = d // H
    for i in range(cfg["enc_layers"]):
        p = f"{prefix}blocks.{i}."
        h = _ln(x, W[p + "attn_ln.weight"], W[p + "attn_ln.bias"])
        q = F.linear(h, W[p + "attn.query.weight"], W[p + "attn.query.bias"])
        k = F.linear(h, W[p + "attn.key.weight"])
        v = F.linear(h, W[p + "attn.value.weight"], W[p + "attn.value.bias"])
        scale = hd ** -0.25
        q = q.view(B, T, H, hd).permute(0, 2, 1, 3) * scale
        k = k.view(B, T, H, hd).permute(0, 2, 3, 1) * scale
        v = v.view(B, T, H, hd).permute(0, 2, 1, 3)
        w = F.softmax((q @ k).float(), dim=-1).to(q.dtype)
        a = (w @ v).permute(0, 2, 1, 3).flatten(start_dim=2)
        x = x + F.linear(a, W[p + "attn.out.weight"], W[p + "attn.out.bias"])
        h = _ln(x, W[p + "mlp_ln.weight"], W[p + "mlp_ln.bias"])
        h = F.linear(F.gelu(F.linear(h, W[p + "mlp.0.weight"], W[p + "mlp.0.bias"])), W[p + "mlp.2.weight"], W[p + "mlp.2.bias"])
        x = x + h
    return _ln(x, W[prefix + "ln_post.weight"], W[prefix + "ln_post.bias"])



# ---------------------------------------------------------------------------------------------- a11: HuBERT encoder
def hubert_config(**kw) -> dict:
    """HuBERT-large geometry (fairseq hubert_large / HF HubertConfig: feat_extract_norm="layer",
    do_stable_layer_norm=True, conv_bias=True); tests shrink the widths, not the structure."""
    c = dict(hub_conv_dim=(512,) * 7, hub_conv_kernel=(10, 3, 3, 3, 3, 2, 2), hub_conv_stride=(5, 2, 2, 2, 2, 2, 2),
             hub_dim=1024, hub_heads=16, hub_layers=24, hub_ffn=4096, hub_pos_k=128, hub_pos_groups=16, hub_eps=1e-5)
    c.update(kw)
    return c


def hubert_base_config(**kw) -> dict:
    """HuBERT-base geometry (fairseq hubert_base: extractor_mode="default", layer_norm_first=False; HF HubertConfig:
    feat_extract_norm="group", do_stable_layer_norm=False, conv_bias=False): GroupNorm over time after the first conv only,
    post-LN layers behind the encoder-level LayerNorm, 12 x 768 / 12 heads / ffn 3072."""
    c = hubert_config(hub_dim=768, hub_heads=12, hub_layers=12, hub_ffn=3072, hub_extractor_mode="default", hub_layer_norm_first=False)
    c.update(kw)
    return c


def hubert_frame_padding_mask(n_samples: int, n_frames: int, n_valid: torch.Tensor) -> torch.Tensor:
    """fairseq HubertModel.forward_padding_mask (fairseq/models/hubert/hubert.py, fairseq is an un-vendored, unpinned dependency
    of the reference: README.md:89-95): the sample-level padding mask [B, N] is cut to a multiple of the frame count, viewed as
    [B, T', N // T'] and a frame is PADDING iff ALL of its samples are -- i.e. clip b keeps ceil(n_valid[b] / (N // T')) frames.
    Returns bool [B, T'] with True = padding (what the reference forwards as `results["padding_mask"]`, slam_model.py:336-341)."""
    chunk = n_samples // n_frames
    keep = torch.clamp((n_valid + chunk - 1) // chunk, max=n_frames)
    return torch.arange(n_frames)[None, :] >= keep[:, None]


def _mul(x, m):
    """multiplicative dropout mask (0 or 1/(1-p)) handed in by the caller, None = identity"""
    return x if m is None else x * m


def hubert_encoder(W: Dict[str, torch.Tensor], cfg: dict, wav: torch.Tensor, prefix="encoder.",
                   n_valid: Optional[torch.Tensor] = None, train: Optional[dict] = None) -> torch.Tensor:
    """The HuBERT branch of slam_model.forward (src/slam_llm/models/slam_model.py:335-341: fairseq
    `self.encoder(source=audio, padding_mask=...)["encoder_out"]`), restated from the HF twin of fairseq's model
    (transformers/models/hubert/modeling_hubert.py: HubertFeatureEncoder with LayerNorm conv layers :127-151,
    HubertFeatureProjection :216-233, HubertPositionalConvEmbedding :45-93 (weight-norm folded into the weight),
    HubertEncoderStableLayerNorm :550-625).
    wav [B, N] (already layer-normed by the dataset, speech_dataset.py:96-97) -> [B, T', hub_dim].
    n_valid [B] (ragged batch, zero-padded waveforms; the reference passes `padding_mask = 1 - audio_mask`): the conv stack runs
    over the padded waveform, the frame mask follows fairseq (hubert_frame_padding_mask), padded frames are zeroed before the
    positional conv (fairseq TransformerEncoder.extract_features: `x = index_put(x, padding_mask, 0)`; HF :571-575 the same) and
    masked as attention KEYS in every layer; their own output rows are garbage (never read: the splice takes the clip's first
    len//320//5 projector frames).
    train (un-frozen encoder in train mode; fairseq wav2vec2 TransformerEncoder / TransformerSentenceEncoderLayer -- the module tree the
    reference's WavLM.py vendors, see wavlm_encoder): the masks the regularisers drew, as inputs (torch's RNG stream cannot be shared
    with a device kernel): {"input": after the feature projection, "x": after the positional conv (+ the encoder LayerNorm of post-LN
    models), "layers": per layer None (dropped by layerdrop) or {"attn": [B,H,T,T] on the attention probabilities, "d1" / "d3": on the
    attention / feed-forward output projections before the residual add, "d2": after the GELU}}; each mask 0 or 1/(1-p), None = off."""
    tr = train or {}
    x = wav[:, None, :]
    group_mode = cfg.get("hub_extractor_mode", "layer_norm") == "default"
    pre_ln = cfg.get("hub_layer_norm_first", True)
    for i, (k, st) in enumerate(zip(cfg["hub_conv_kernel"], cfg["hub_conv_stride"])):
        p = f"{prefix}feature_extractor.conv_layers.{i}."
        x = F.conv1d(x, W[p + "conv.weight"], W.get(p + "conv.bias"), stride=st)
        if group_mode:      # HubertGroupNormConvLayer for layer 0 (modeling_hubert.py:153-176), HubertNoLayerNormConvLayer after it
            if i == 0:
                x = F.group_norm(x, x.shape[1], W[p + "layer_norm.weight"], W[p + "layer_norm.bias"], 1e-5)
            x = F.gelu(x)
        else:
            x = F.layer_norm(x.transpose(-2, -1), (x.shape[1],), W[p + "layer_norm.weight"], W[p + "layer_norm.bias"], 1e-5)
            x = F.gelu(x.transpose(-2, -1))
    x = x.transpose(1, 2)  # [B, T', C]
    eps = cfg["hub_eps"]
    p = prefix + "feature_projection."
    x = F.layer_norm(x, (x.shape[-1],), W[p + "layer_norm.weight"], W[p + "layer_norm.bias"], eps)
    x = _mul(F.linear(x, W[p + "projection.weight"], W[p + "projection.bias"]), tr.get("input"))
    p = prefix + "encoder."
    kpos = cfg["hub_pos_k"]
    key_bias = None
    if n_valid is not None:
        pad = hubert_frame_padding_mask(wav.shape[1], x.shape[1], n_valid)
        x = x.masked_fill(pad[:, :, None], 0.0)
        key_bias = torch.zeros(pad.shape).masked_fill(pad, float("-inf"))[:, None, None, :]
    pw = W.get(p + "pos_conv_embed.conv.weight")
    if pw is None:      # weight_norm(dim=2) kept as parameters (HF: parametrizations.weight.original0 = g [1, 1, k], original1 = v;
        g_, v_ = W[p + "pos_conv_embed.conv.parametrizations.weight.original0"], W[p + "pos_conv_embed.conv.parametrizations.weight.original1"]
        pw = g_ * v_ / v_.norm(dim=(0, 1), keepdim=True)   # fairseq: encoder.pos_conv.0.weight_g / weight_v -- what the reference trains)
    pos = F.conv1d(x.transpose(1, 2), pw, W[p + "pos_conv_embed.conv.bias"], padding=kpos // 2, groups=cfg["hub_pos_groups"])
    if kpos % 2 == 0:
        pos = pos[:, :, :-1]
    x = x + F.gelu(pos).transpose(1, 2)
    B, T, d = x.shape
    H = cfg["hub_heads"]
    hd = d // H
    if not pre_ln:      # HubertEncoder (post-LN, modeling_hubert.py:470-520): the encoder LayerNorm precedes the layers
        x = F.layer_norm(x, (d,), W[p + "layer_norm.weight"], W[p + "layer_norm.bias"], eps)
    x = _mul(x, tr.get("x"))
    for i in range(cfg["hub_layers"]):
        q_ = f"{p}layers.{i}."
        lm = tr["layers"][i] if "layers" in tr else {}
        if lm is None:      # layerdrop
            continue
        h = F.layer_norm(x, (d,), W[q_ + "layer_norm.weight"], W[q_ + "layer_norm.bias"], eps) if pre_ln else x
        q = F.linear(h, W[q_ + "attention.q_proj.weight"], W[q_ + "attention.q_proj.bias"]).view(B, T, H, hd).transpose(1, 2)
        k = F.linear(h, W[q_ + "attention.k_proj.weight"], W[q_ + "attention.k_proj.bias"]).view(B, T, H, hd).transpose(1, 2)
        v = F.linear(h, W[q_ + "attention.v_proj.weight"], W[q_ + "attention.v_proj.bias"]).view(B, T, H, hd).transpose(1, 2)
        sc = (q @ k.transpose(2, 3)) * hd ** -0.5
        if key_bias is not None:
            sc = sc + key_bias
        a = _mul(F.softmax(sc, dim=-1), lm.get("attn")) @ v
        a = a.transpose(1, 2).reshape(B, T, d)
        x = x + _mul(F.linear(a, W[q_ + "attention.out_proj.weight"], W[q_ + "attention.out_proj.bias"]), lm.get("d1"))
        if not pre_ln:  # HubertEncoderLayer (:395-420): x = LN(x + attn(x)); x = LN_final(x + ffn(x))
            x = F.layer_norm(x, (d,), W[q_ + "layer_norm.weight"], W[q_ + "layer_norm.bias"], eps)
        h = F.layer_norm(x, (d,), W[q_ + "final_layer_norm.weight"], W[q_ + "final_layer_norm.bias"], eps) if pre_ln else x
        h = F.linear(_mul(F.gelu(F.linear(h, W[q_ + "feed_forward.intermediate_dense.weight"], W[q_ + "feed_forward.intermediate_dense.bias"])), lm.get("d2")),
                     W[q_ + "feed_forward.output_dense.weight"], W[q_ + "feed_forward.output_dense.bias"])
        x = x + _mul(h, lm.get("d3"))
        if not pre_ln:
            x = F.layer_norm(x, (d,), W[q_ + "final_layer_norm.weight"], W[q_ + "final_layer_norm.bias"], eps)
    return F.layer_norm(x, (d,), W[p + "layer_norm.weight"], W[p + "layer_norm.bias"], eps) if pre_ln else x


def init_hubert_weights(cfg: dict, seed: int = 7, prefix="encoder.", weight_norm: bool = False) -> Dict[str, torch.Tensor]:
    """weight_norm=True: the positional conv keeps its (g, v) parametrisation (g != ||v||) instead of the folded weight"""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    W = {}
    cin = 1
    for i, (co, k) in enumerate(zip(cfg["hub_conv_dim"], cfg["hub_conv_kernel"])):
        p = f"{prefix}feature_extractor.conv_layers.{i}."
        W[p + "conv.weight"] = rn(co, cin, k, std=(1.0 / (cin * k)) ** 0.5)
        if cfg.get("hub_extractor_mode", "layer_norm") == "default":     # base: no conv bias, GroupNorm on layer 0 only
            if i == 0:
                W[p + "layer_norm.weight"], W[p + "layer_norm.bias"] = 1 + rn(co, std=0.1), rn(co, std=0.1)
        else:
            W[p + "conv.bias"] = rn(co)
            W[p + "layer_norm.weight"] = 1 + rn(co, std=0.1)
            W[p + "layer_norm.bias"] = rn(co, std=0.1)
        cin = co
    d, ffn = cfg["hub_dim"], cfg["hub_ffn"]
    p = prefix + "feature_projection."
    W[p + "layer_norm.weight"], W[p + "layer_norm.bias"] = 1 + rn(cin, std=0.1), rn(cin, std=0.1)
    W[p + "projection.weight"], W[p + "projection.bias"] = rn(d, cin, std=cin ** -0.5), rn(d)
    p = prefix + "encoder."
    gch = d // cfg["hub_pos_groups"]
    W[p + "pos_conv_embed.conv.weight"] = rn(d, gch, cfg["hub_pos_k"], std=(1.0 / (gch * cfg["hub_pos_k"])) ** 0.5)
    W[p + "pos_conv_embed.conv.bias"] = rn(d)
    if weight_norm:
        v_ = W.pop(p + "pos_conv_embed.conv.weight")
        W[p + "pos_conv_embed.conv.parametrizations.weight.original1"] = v_
        W[p + "pos_conv_embed.conv.parametrizations.weight.original0"] = v_.norm(dim=(0, 1), keepdim=True) * (1 + rn(1, 1, cfg["hub_pos_k"], std=0.2))
    for i in range(cfg["hub_layers"]):
        q_ = f"{p}layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            W[q_ + f"attention.{n}.weight"], W[q_ + f"attention.{n}.bias"] = rn(d, d, std=d ** -0.5), rn(d)
        W[q_ + "layer_norm.weight"], W[q_ + "layer_norm.bias"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
        W[q_ + "final_layer_norm.weight"], W[q_ + "final_layer_norm.bias"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
        W[q_ + "feed_forward.intermediate_dense.weight"], W[q_ + "feed_forward.intermediate_dense.bias"] = rn(ffn, d, std=d ** -0.5), rn(ffn)
        W[q_ + "feed_forward.output_dense.weight"], W[q_ + "feed_forward.output_dense.bias"] = rn(d, ffn, std=ffn ** -0.5), rn(d)
    W[p + "layer_norm.weight"], W[p + "layer_norm.bias"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
    return W

# ---------------------------------------------------------------------------------------------- f4: WavLM encoder
def wavlm_config(**kw) -> dict:
    """WavLM-Large geometry (src/slam_llm/models/wavlm/WavLM.py:162-214 fields as the released checkpoint sets them:
    extractor_mode="layer_norm", conv_bias=False, normalize=True, layer_norm_first=True, 24 x 1024 / 16 heads / ffn 4096,
    conv_pos 128 / 16 groups, relative_position_embedding + gru_rel_pos with 320 buckets, max_distance 800).  The conv stack,
    feature projection and positional conv share HuBERT's keys (hub_*); tests shrink widths, not structure."""
    c = hubert_config()
    c.update(wavlm_buckets=320, wavlm_max_distance=800, hub_extractor_mode="layer_norm", hub_layer_norm_first=True)
    c.update(kw)
    return c


def wavlm_base_config(**kw) -> dict:
    """WavLM Base / Base+ geometry (the released checkpoints' cfg): extractor_mode="default" (GroupNorm after the first conv only),
    layer_norm_first=False (post-LN layers), 12 x 768 / 12 heads / ffn 3072; everything else as Large."""
    c = wavlm_config(hub_dim=768, hub_heads=12, hub_layers=12, hub_ffn=3072, hub_extractor_mode="default", hub_layer_norm_first=False)
    c.update(kw)
    return c


def wavlm_relative_buckets(rel: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """MultiheadAttention._relative_positions_bucket, bidirectional (src/slam_llm/models/wavlm/modules.py:417-442): half of the
    buckets per sign; exact up to num_buckets/4, then log-spaced up to max_distance, clamped to the last bucket."""
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    a = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(a.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(a < max_exact, a, large)


def wavlm_encoder(W: Dict[str, torch.Tensor], cfg: dict, wav: torch.Tensor, prefix="encoder.model.",
                  n_valid: Optional[torch.Tensor] = None, train: Optional[dict] = None) -> torch.Tensor:
    """The WavLM branch of slam_model.forward (src/slam_llm/models/slam_model.py:333-334:
    `self.encoder.extract_features(audio, 1 - audio_mask)` -> models/encoder.py:126-127 -> WavLM.extract_features,
    models/wavlm/WavLM.py:323-376) in eval mode.  wav [B, N] (layer-normed by the dataset when cfg.normalize) -> [B, T', d].
    * feature extractor, "layer_norm" mode (WavLM.py:378-505): conv (no bias) -> LayerNorm over channels -> GELU, 7 times;
    * LayerNorm -> post_extract_proj (:343-351); padding mask per frame = all samples of the frame padded (:311-321);
    * TransformerEncoder.extract_features (:572-613): padded frames zeroed, x += GELU(SamePad(weight-normed grouped conv(x)));
    * layer_norm_first layers (:690-715): x += attn(LN(x)); x += fc2(gelu(fc1(LN(x)))); final LayerNorm (:567-568);
    * attention (modules.py:504-562): scores = q.k / sqrt(hd) + gate[b,h,q] * position_bias[h,q,k], where position_bias is
      layer 0's `relative_attention_bias` embedding of the bucketed distance k - q (compute_bias :444-455, shared by ALL
      layers) and gate = a * (g * grep_a - 1) + 2 with (a, g) = sigmoid of the two 4-sums of grep_linear(per-head slice of the
      attention INPUT) (:522-531); key padding mask -> -inf.
    train (the module is left in train mode when un-frozen, slam_model.py:317-318): the masks of WavLM's regularisers as inputs, same
    dict as hubert_encoder -- "input" = dropout_input (:353), "x" = F.dropout after the positional conv (+ LayerNorm) (:582-584),
    per layer None = skipped by layerdrop (:596-597) or "attn" (attention_dropout on the probabilities, modules.py F.dropout on
    attn_probs), "d1" / "d2" / "d3" (:702-726).  With layer 0 skipped, position_bias is never created (it is computed inside layer 0's
    attention and handed down, :593-599): the other layers then run without the bias and the gate."""
    tr = train or {}
    x = wav[:, None, :]
    group_mode = cfg.get("hub_extractor_mode", "layer_norm") == "default"
    pre_ln = cfg.get("hub_layer_norm_first", True)
    for i, (k, st) in enumerate(zip(cfg["hub_conv_kernel"], cfg["hub_conv_stride"])):
        p = f"{prefix}feature_extractor.conv_layers.{i}."
        x = F.conv1d(x, W[p + "0.weight"], None, stride=st)
        if group_mode:
            # extractor_mode "default" (WavLM.py:428-441, Base / Base+): Fp32GroupNorm(dim, dim) -- one group per channel, i.e.
            # each channel normalised over TIME -- after the first conv only; the other layers are conv -> GELU
            if i == 0:
                x = F.group_norm(x, x.shape[1], W[p + "2.weight"], W[p + "2.bias"], 1e-5)
            x = F.gelu(x)
        else:
            x = F.layer_norm(x.transpose(-2, -1), (x.shape[1],), W[p + "2.1.weight"], W[p + "2.1.bias"], 1e-5)
            x = F.gelu(x.transpose(-2, -1))
    x = x.transpose(1, 2)
    x = F.layer_norm(x, (x.shape[-1],), W[prefix + "layer_norm.weight"], W[prefix + "layer_norm.bias"], 1e-5)
    x = _mul(F.linear(x, W[prefix + "post_extract_proj.weight"], W[prefix + "post_extract_proj.bias"]), tr.get("input"))
    B, T, d = x.shape
    key_bias = None
    if n_valid is not None:
        pad = hubert_frame_padding_mask(wav.shape[1], T, n_valid)      # WavLM.forward_padding_mask: the same rule
        x = x.masked_fill(pad[:, :, None], 0.0)
        key_bias = torch.zeros(pad.shape).masked_fill(pad, float("-inf"))[:, None, None, :]
    p = prefix + "encoder."
    g_, v_ = W[p + "pos_conv.0.weight_g"], W[p + "pos_conv.0.weight_v"]      # nn.utils.weight_norm(dim=2): w = g * v / ||v||_(0,1)
    pw = g_ * v_ / v_.norm(dim=(0, 1), keepdim=True)
    kpos = cfg["hub_pos_k"]
    pos = F.conv1d(x.transpose(1, 2), pw, W[p + "pos_conv.0.bias"], padding=kpos // 2, groups=cfg["hub_pos_groups"])
    if kpos % 2 == 0:
        pos = pos[:, :, :-1]
    x = x + F.gelu(pos).transpose(1, 2)
    if not pre_ln:      # post-LN encoders normalise HERE (WavLM.py:582-583) and not after the layers (:567-568)
        x = F.layer_norm(x, (d,), W[p + "layer_norm.weight"], W[p + "layer_norm.bias"], 1e-5)
    x = _mul(x, tr.get("x"))
    H = cfg["hub_heads"]
    hd = d // H
    rel = torch.arange(T)[None, :] - torch.arange(T)[:, None]                 # memory - context = k - q
    buckets = wavlm_relative_buckets(rel, cfg["wavlm_buckets"], cfg["wavlm_max_distance"])
    pos_bias = F.embedding(buckets, W[p + "layers.0.self_attn.relative_attention_bias.weight"]).permute(2, 0, 1)   # [H, T, T]
    biased = not ("layers" in tr and tr["layers"][0] is None)
    for i in range(cfg["hub_layers"]):
        q_ = f"{p}layers.{i}."
        lm = tr["layers"][i] if "layers" in tr else {}
        if lm is None:
            continue
        # attention input: LN(x) in layer_norm_first layers (:690-703), x itself in post-LN layers (:716-725)
        h = F.layer_norm(x, (d,), W[q_ + "self_attn_layer_norm.weight"], W[q_ + "self_attn_layer_norm.bias"], 1e-5) if pre_ln else x
        hh = h.view(B, T, H, hd).permute(0, 2, 1, 3)                           # per-head slices of the attention input
        gl = F.linear(hh, W[q_ + "self_attn.grep_linear.weight"], W[q_ + "self_attn.grep_linear.bias"]).view(B, H, T, 2, 4).sum(-1)
        ga, gb = torch.sigmoid(gl).chunk(2, dim=-1)
        gate = ga * (gb * W[q_ + "self_attn.grep_a"] - 1.0) + 2.0              # [B, H, T, 1]
        q = F.linear(h, W[q_ + "self_attn.q_proj.weight"], W[q_ + "self_attn.q_proj.bias"]).view(B, T, H, hd).transpose(1, 2)
        k = F.linear(h, W[q_ + "self_attn.k_proj.weight"], W[q_ + "self_attn.k_proj.bias"]).view(B, T, H, hd).transpose(1, 2)
        v = F.linear(h, W[q_ + "self_attn.v_proj.weight"], W[q_ + "self_attn.v_proj.bias"]).view(B, T, H, hd).transpose(1, 2)
        sc = (q @ k.transpose(2, 3)) * hd ** -0.5
        if biased:
            sc = sc + gate * pos_bias[None]
        if key_bias is not None:
            sc = sc + key_bias
        a = (_mul(F.softmax(sc, dim=-1), lm.get("attn")) @ v).transpose(1, 2).reshape(B, T, d)
        x = x + _mul(F.linear(a, W[q_ + "self_attn.out_proj.weight"], W[q_ + "self_attn.out_proj.bias"]), lm.get("d1"))
        ffn = lambda t: _mul(F.linear(_mul(F.gelu(F.linear(t, W[q_ + "fc1.weight"], W[q_ + "fc1.bias"])), lm.get("d2")),    # noqa: E731
                                      W[q_ + "fc2.weight"], W[q_ + "fc2.bias"]), lm.get("d3"))
        if pre_ln:
            h = F.layer_norm(x, (d,), W[q_ + "final_layer_norm.weight"], W[q_ + "final_layer_norm.bias"], 1e-5)
            x = x + ffn(h)
        else:           # post-LN (:726-739): x = LN1(x + attn(x)); x = LN2(x + ffn(x))
            x = F.layer_norm(x, (d,), W[q_ + "self_attn_layer_norm.weight"], W[q_ + "self_attn_layer_norm.bias"], 1e-5)
            x = x + ffn(x)
            x = F.layer_norm(x, (d,), W[q_ + "final_layer_norm.weight"], W[q_ + "final_layer_norm.bias"], 1e-5)
    return F.layer_norm(x, (d,), W[p + "layer_norm.weight"], W[p + "layer_norm.bias"], 1e-5) if pre_ln else x


def init_wavlm_weights(cfg: dict, seed: int = 9, prefix="encoder.model.") -> Dict[str, torch.Tensor]:
    """seeded weights under the reference WavLM's own state-dict names (as `slam_model.state_dict()` would carry them:
    `encoder` = WavLMEncoder, `.model` = WavLM, models/encoder.py:109-127)."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    W = {}
    cin = 1
    for i, (co, k) in enumerate(zip(cfg["hub_conv_dim"], cfg["hub_conv_kernel"])):
        p = f"{prefix}feature_extractor.conv_layers.{i}."
        W[p + "0.weight"] = rn(co, cin, k, std=(1.0 / (cin * k)) ** 0.5)
        if cfg.get("hub_extractor_mode", "layer_norm") == "default":
            if i == 0:
                W[p + "2.weight"], W[p + "2.bias"] = 1 + rn(co, std=0.1), rn(co, std=0.1)
        else:
            W[p + "2.1.weight"], W[p + "2.1.bias"] = 1 + rn(co, std=0.1), rn(co, std=0.1)
        cin = co
    d, ffn, H = cfg["hub_dim"], cfg["hub_ffn"], cfg["hub_heads"]
    W[prefix + "layer_norm.weight"], W[prefix + "layer_norm.bias"] = 1 + rn(cin, std=0.1), rn(cin, std=0.1)
    W[prefix + "post_extract_proj.weight"], W[prefix + "post_extract_proj.bias"] = rn(d, cin, std=cin ** -0.5), rn(d)
    W[prefix + "mask_emb"] = torch.rand(d, generator=g)
    p = prefix + "encoder."
    gch = d // cfg["hub_pos_groups"]
    W[p + "pos_conv.0.weight_v"] = rn(d, gch, cfg["hub_pos_k"], std=(1.0 / (gch * cfg["hub_pos_k"])) ** 0.5)
    W[p + "pos_conv.0.weight_g"] = W[p + "pos_conv.0.weight_v"].norm(dim=(0, 1), keepdim=True) * (1 + rn(1, 1, cfg["hub_pos_k"], std=0.1))
    W[p + "pos_conv.0.bias"] = rn(d)
    for i in range(cfg["hub_layers"]):
        q_ = f"{p}layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            W[q_ + f"self_attn.{n}.weight"], W[q_ + f"self_attn.{n}.bias"] = rn(d, d, std=d ** -0.5), rn(d)
        W[q_ + "self_attn.grep_linear.weight"], W[q_ + "self_attn.grep_linear.bias"] = rn(8, d // H, std=0.3), rn(8, std=0.3)
        W[q_ + "self_attn.grep_a"] = 1 + rn(1, H, 1, 1, std=0.3)
        if i == 0:
            W[q_ + "self_attn.relative_attention_bias.weight"] = rn(cfg["wavlm_buckets"], H, std=1.0)
        W[q_ + "self_attn_layer_norm.weight"], W[q_ + "self_attn_layer_norm.bias"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
        W[q_ + "fc1.weight"], W[q_ + "fc1.bias"] = rn(ffn, d, std=d ** -0.5), rn(ffn)
        W[q_ + "fc2.weight"], W[q_ + "fc2.bias"] = rn(d, ffn, std=ffn ** -0.5), rn(d)
        W[q_ + "final_layer_norm.weight"], W[q_ + "final_layer_norm.bias"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
    W[p + "layer_norm.weight"], W[p + "layer_norm.bias"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
    return W


# ---------------------------------------------------------------------------------------------- a3: projector
def projector_concat(W, x: torch.Tensor, k: int, prefix="encoder_projector.") -> torch.Tensor:
    """EncoderProjectorConcat.forward, src/slam_llm/models/projector.py:15-27."""
    B, T, d = x.shape
    drop = T % k
    if drop > 0:
        x = x[:, :-drop, :]
    x = x.contiguous().view(B, x.shape[1] // k, d * k)
    x = F.relu(F.linear(x, W[prefix + "linear1.weight"], W[prefix + "linear1.bias"]))
    return F.linear(x, W[prefix + "linear2.weight"], W[prefix + "linear2.bias"])



def projector_cov1d(W, x: torch.Tensor, k: int, prefix="encoder_projector.") -> torch.Tensor:
    """EncoderProjectorCov1d.forward, src/slam_llm/models/projector.py:29-49: Conv1d(d, d, kernel=k, stride=k) over
    time, ReLU, Linear(d, 2048), ReLU, Linear(2048, llm_dim).  kernel = stride and no padding, so the conv is a
    linear map of each k-frame stack: out[t] = sum_j W[:, :, j] x[t*k + j] + b (tail frames beyond k*(T//k) dropped)."""
    B, T, d = x.shape
    Ta = T // k
    frames = x[:, : Ta * k].reshape(B, Ta, k, d)
    c = torch.einsum("btjc,ocj->bto", frames, W[prefix + "conv1d.weight"]) + W[prefix + "conv1d.bias"]
    h = F.relu(F.linear(F.relu(c), W[prefix + "linear1.weight"], W[prefix + "linear1.bias"]))
    return F.linear(h, W[prefix + "linear2.weight"], W[prefix + "linear2.bias"])


def init_cov1d_weights(enc_dim: int, llm_dim: int, k: int, hidden: int = 2048, seed: int = 13,
                       prefix="encoder_projector.") -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, std: torch.randn(*s, generator=g) * std  # noqa: E731
    return {prefix + "conv1d.weight": rn(enc_dim, enc_dim, k, std=(enc_dim * k) ** -0.5),
            prefix + "conv1d.bias": rn(enc_dim, std=0.05),
            prefix + "linear1.weight": rn(hidden, enc_dim, std=enc_dim ** -0.5),
            prefix + "linear1.bias": rn(hidden, std=0.05),
            prefix + "linear2.weight": rn(llm_dim, hidden, std=hidden ** -0.5),
            prefix + "linear2.bias": rn(llm_dim, std=0.05)}


# ---------------------------------------------------------------------------------------------- a3': Q-Former projector
def qformer_config(**kw) -> dict:
    """Blip2QFormerConfig defaults as used by EncoderProjectorQFormer (src/slam_llm/models/projector.py:52-67):
    hidden 768, 12 heads, ffn 3072, cross-attention every 2nd layer, LayerNorm eps 1e-12; layers/queries from the recipe."""
    c = dict(qf_dim=768, qf_heads=12, qf_ffn=3072, qf_layers=8, qf_queries=64, qf_eps=1e-12, qf_cross_freq=2)
    c.update(kw)
    return c


def _mha(W, p, hq, hkv, H, key_mask=None, prob_mask=None):
    B, Tq, d = hq.shape
    Tk = hkv.shape[1]
    hd = d // H
    q = F.linear(hq, W[p + "query.weight"], W[p + "query.bias"]).view(B, Tq, H, hd).transpose(1, 2)
    k = F.linear(hkv, W[p + "key.weight"], W[p + "key.bias"]).view(B, Tk, H, hd).transpose(1, 2)
    v = F.linear(hkv, W[p + "value.weight"], W[p + "value.bias"]).view(B, Tk, H, hd).transpose(1, 2)
    s = (q @ k.transpose(2, 3)) * hd ** -0.5
    if key_mask is not None:
        s = s.masked_fill(~key_mask.bool()[:, None, None, :], torch.finfo(torch.float32).min)
    pr = F.softmax(s, dim=-1)
    if prob_mask is not None:     # attention_probs_dropout_prob (train mode): [B, H, Tq, Tk], values 0 or 1/(1-p)
        pr = pr * prob_mask
    return (pr @ v).transpose(1, 2).reshape(B, Tq, d)


def projector_qformer(W, cfg, x: torch.Tensor, atts: Optional[torch.Tensor], prefix="encoder_projector.",
                      hidden_masks: Optional[list] = None, attn_masks: Optional[list] = None) -> torch.Tensor:
    """EncoderProjectorQFormer.forward (src/slam_llm/models/projector.py:69-80) over HF Blip2QFormerModel
    (transformers/models/blip_2/modeling_blip_2.py:536-760, 849-940).  x [B, Tk, d_enc], atts [B, Tk] (1 = attend) ->
    [B, Q, llm_dim].  hidden_masks=None is eval mode (no dropout).  Train mode: the stack's hidden dropouts
    (hidden_dropout_prob: after the query LayerNorm, Blip2QFormerModel.forward; after each output projection before the
    residual add, Blip2QFormerSelfOutput / Blip2QFormerOutput) multiply by the given masks [B, Q, d] (values 0 or 1/(1-p)),
    consumed in forward order -- torch's dropout RNG stream cannot be shared with a device kernel, so the masks are an input.
    attn_masks: the attention-probability dropout (attention_probs_dropout_prob, Blip2QFormerMultiHeadAttention: applied to the
    softmax output before the value product), one [B, H, Tq, Tk] mask per attention call in forward order (self, cross, ...)."""
    eps, H = cfg["qf_eps"], cfg["qf_heads"]
    B = x.shape[0]
    d = cfg["qf_dim"]
    P = prefix + "qformer."
    masks = iter(hidden_masks) if hidden_masks is not None else None
    amasks = iter(attn_masks) if attn_masks is not None else None

    def drop(t):
        return t if masks is None else t * next(masks)

    def amask():
        return None if amasks is None else next(amasks)

    h = drop(F.layer_norm(W[prefix + "query"].expand(B, -1, -1), (d,), W[P + "layernorm.weight"], W[P + "layernorm.bias"], eps))
    for l in range(cfg["qf_layers"]):
        L = f"{P}encoder.layer.{l}."
        a = _mha(W, L + "attention.attention.", h, h, H, prob_mask=amask())
        h = F.layer_norm(drop(F.linear(a, W[L + "attention.output.dense.weight"], W[L + "attention.output.dense.bias"])) + h, (d,),
                         W[L + "attention.output.LayerNorm.weight"], W[L + "attention.output.LayerNorm.bias"], eps)
        if l % cfg["qf_cross_freq"] == 0:
            c = _mha(W, L + "crossattention.attention.", h, x, H, atts, prob_mask=amask())
            h = F.layer_norm(drop(F.linear(c, W[L + "crossattention.output.dense.weight"], W[L + "crossattention.output.dense.bias"])) + h,
                             (d,), W[L + "crossattention.output.LayerNorm.weight"], W[L + "crossattention.output.LayerNorm.bias"], eps)
        f = F.gelu(F.linear(h, W[L + "intermediate_query.dense.weight"], W[L + "intermediate_query.dense.bias"]))
        h = F.layer_norm(drop(F.linear(f, W[L + "output_query.dense.weight"], W[L + "output_query.dense.bias"])) + h, (d,),
                         W[L + "output_query.LayerNorm.weight"], W[L + "output_query.LayerNorm.bias"], eps)
    y = F.linear(h, W[prefix + "linear.weight"], W[prefix + "linear.bias"])
    return F.layer_norm(y, (y.shape[-1],), W[prefix + "norm.weight"], W[prefix + "norm.bias"], 1e-5)


def init_qformer_weights(cfg: dict, enc_dim: int, llm_dim: int, seed: int = 11, prefix="encoder_projector.") -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    d, ffn = cfg["qf_dim"], cfg["qf_ffn"]
    W = {prefix + "query": rn(1, cfg["qf_queries"], d, std=1.0)}
    P = prefix + "qformer."
    W[P + "layernorm.weight"], W[P + "layernorm.bias"] = 1 + rn(d, std=0.1), rn(d, std=0.1)

    def attn(p, kv_dim):
        for n, kin in (("query", d), ("key", kv_dim), ("value", kv_dim)):
            W[p + f"attention.{n}.weight"], W[p + f"attention.{n}.bias"] = rn(d, kin, std=kin ** -0.5), rn(d)
        W[p + "output.dense.weight"], W[p + "output.dense.bias"] = rn(d, d, std=d ** -0.5), rn(d)
        W[p + "output.LayerNorm.weight"], W[p + "output.LayerNorm.bias"] = 1 + rn(d, std=0.1), rn(d, std=0.1)

    for l in range(cfg["qf_layers"]):
        L = f"{P}encoder.layer.{l}."
        attn(L + "attention.", d)
        if l % cfg["qf_cross_freq"] == 0:
            attn(L + "crossattention.", enc_dim)
        W[L + "intermediate_query.dense.weight"], W[L + "intermediate_query.dense.bias"] = rn(ffn, d, std=d ** -0.5), rn(ffn)
        W[L + "output_query.dense.weight"], W[L + "output_query.dense.bias"] = rn(d, ffn, std=ffn ** -0.5), rn(d)
        W[L + "output_query.LayerNorm.weight"], W[L + "output_query.LayerNorm.bias"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
    W[prefix + "linear.weight"], W[prefix + "linear.bias"] = rn(llm_dim, d, std=d ** -0.5), rn(llm_dim)
    W[prefix + "norm.weight"], W[prefix + "norm.bias"] = 1 + rn(llm_dim, std=0.1), rn(llm_dim, std=0.1)
    return W

# ---------------------------------------------------------------------------------------------- a4: embed + splice
def embed_splice(embed_weight, input_ids, modality_mask, encoder_outs):
    """src/slam_llm/models/slam_model.py:370-392 (input_ids is mutated in place like the reference)."""
    input_ids[input_ids == -1] = 0
    inputs_embeds = F.embedding(input_ids, embed_weight)
    start = (modality_mask == True).float().argmax(dim=1)  # noqa: E712
    lengths = torch.clamp(modality_mask.sum(dim=1), max=encoder_outs.shape[1]).tolist()
    pad = torch.zeros_like(inputs_embeds)
    for i in range(encoder_outs.shape[0]):
        pad[i, start[i]: start[i] + lengths[i]] = encoder_outs[i][: lengths[i]]
    return pad + inputs_embeds * (~modality_mask[:, :, None])


# ---------------------------------------------------------------------------------------------- a5/a6: llama + LoRA
def _rmsnorm(x, w, eps):
    dt = x.dtype
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(v + eps)).to(dt)


def rope_tables(T: int, D: int, theta: float):
    inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    freqs = torch.arange(T, dtype=torch.float32)[:, None] * inv[None, :]
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos(), emb.sin()


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def lora_linear(W, name: str, x: torch.Tensor, cfg: dict) -> torch.Tensor:
    """peft 0.6.0 LoRA Linear: F.linear(x, W) + lora_B(lora_A(dropout(x))) * (alpha / r); dropout = 0 for parity."""
    y = F.linear(x, W[name + ".weight"])
    a = W.get(name + ".lora_A.default.weight")
    if a is not None:
        b = W[name + ".lora_B.default.weight"]
        y = y + F.linear(F.linear(x, a), b) * (cfg["lora_alpha"] / cfg["lora_r"])
    return y


def llama_forward(W, cfg, inputs_embeds, attention_mask, labels=None, prefix="llm.base_model.model.",
                  position_ids=None):
    """HF LlamaForCausalLM(inputs_embeds, attention_mask, labels) as called at slam_model.py:400
    (transformers/models/llama/modeling_llama.py; loss transformers/loss/loss_utils.py:32-70).
    positions = arange(T) for every row unless position_ids [B,T] is given (the generate() path:
    HF derives them from the mask); mask = causal ^ key padding, additive finfo.min."""
    B, T, d = inputs_embeds.shape
    Hq, Hkv, D = cfg["llm_heads"], cfg["llm_kv_heads"], cfg["llm_head_dim"]
    eps = cfg["rms_eps"]
    cos, sin = rope_tables(T, D, cfg["rope_theta"])
    if position_ids is not None:
        cos, sin = cos[position_ids][:, None], sin[position_ids][:, None]  # [B,1,T,D]
    minv = torch.finfo(torch.float32).min
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
    allowed = causal[None, None] & attention_mask.bool()[:, None, None, :]
    add_mask = torch.zeros(B, 1, T, T).masked_fill(~allowed, minv)
    h = inputs_embeds
    for i in range(cfg["llm_layers"]):
        p = f"{prefix}model.layers.{i}."
        res = h
        x = _rmsnorm(h, W[p + "input_layernorm.weight"], eps)
        q = lora_linear(W, p + "self_attn.q_proj", x, cfg).view(B, T, Hq, D).transpose(1, 2)
        k = lora_linear(W, p + "self_attn.k_proj", x, cfg).view(B, T, Hkv, D).transpose(1, 2)
        v = lora_linear(W, p + "self_attn.v_proj", x, cfg).view(B, T, Hkv, D).transpose(1, 2)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        rep = Hq // Hkv
        k = k[:, :, None].expand(B, Hkv, rep, T, D).reshape(B, Hq, T, D)
        v = v[:, :, None].expand(B, Hkv, rep, T, D).reshape(B, Hq, T, D)
        att = (q @ k.transpose(2, 3)) * (D ** -0.5) + add_mask
        att = F.softmax(att, dim=-1, dtype=torch.float32)
        o = (att @ v).transpose(1, 2).reshape(B, T, Hq * D)
        h = res + lora_linear(W, p + "self_attn.o_proj", o, cfg)
        res = h
        x = _rmsnorm(h, W[p + "post_attention_layernorm.weight"], eps)
        g = lora_linear(W, p + "mlp.gate_proj", x, cfg)
        u = lora_linear(W, p + "mlp.up_proj", x, cfg)
        h = res + lora_linear(W, p + "mlp.down_proj", F.silu(g) * u, cfg)
    h = _rmsnorm(h, W[prefix + "model.norm.weight"], eps)
    logits = F.linear(h, W[prefix + "lm_head.weight"])
    loss = None
    if labels is not None:
        lg = logits.float()
        sl = F.pad(labels, (0, 1), value=-100)[..., 1:].contiguous()
        loss = F.cross_entropy(lg.view(-1, lg.shape[-1]), sl.view(-1), ignore_index=-100, reduction="mean")
    return loss, logits


def compute_accuracy(pad_outputs, pad_targets, ignore_label=-100):
    """src/slam_llm/utils/metric.py:3-19."""
    mask = pad_targets != ignore_label
    num = torch.sum(pad_outputs.masked_select(mask) == pad_targets.masked_select(mask))
    return num.float() / torch.sum(mask).float()


def slam_forward(W, cfg, batch: dict):
    """slam_model.forward, src/slam_llm/models/slam_model.py:283-407, whisper + linear projector branch.
    Returns (loss, logits, acc, aux dict of intermediates)."""
    mel = batch["audio_mel"]
    enc = whisper_encoder(W, cfg, mel.permute(0, 2, 1))
    proj = projector_concat(W, enc, cfg["ds_rate"])
    ids = batch["input_ids"].clone()
    emb_w = W["llm.base_model.model.model.embed_tokens.weight"]
    embeds = embed_splice(emb_w, ids, batch["modality_mask"].bool(), proj)
    loss, logits = llama_forward(W, cfg, embeds, batch["attention_mask"], batch.get("labels"))
    acc = None
    if batch.get("labels") is not None:
        preds = torch.argmax(logits, -1)
        acc = compute_accuracy(preds[:, :-1], batch["labels"][:, 1:], -100)
    return loss, logits, acc, {"encoder_out": enc, "projector_out": proj, "inputs_embeds": embeds}


TRAINABLE_MARKERS = ("encoder_projector.", "lora_A", "lora_B")


def trainable_names(W) -> List[str]:
    return [n for n in W if any(m in n for m in TRAINABLE_MARKERS)]


def lr_lambda(step: int, warmup: int, total: int) -> float:
    """src/slam_llm/pipeline/finetune.py:253-260."""
    if step < warmup:
        return min(step / warmup, 1)
    return max(0.0, 1 - (step - warmup) / (total - warmup))


def train_steps(W, cfg, batches: List[dict], lr=1e-4, weight_decay=0.0, warmup=1000, total=100000, train_encoder=False):
    """Loop body of src/slam_llm/utils/train_utils.py:112-169 (fp32, no autocast, grad-accum 1) with
    torch.optim.AdamW + LambdaLR as built at pipeline/finetune.py:247-260.  Mutates W in place.
    train_encoder: train_config.freeze_encoder=false (models/slam_model.py:110-113) -- every encoder PARAMETER trains too
    (openai-whisper's positional_embedding is a registered buffer, not a parameter)."""
    names = trainable_names(W)
    if train_encoder:
        names = names + [n for n in W if n.startswith("encoder.") and not n.endswith("positional_embedding")]
    for n in W:
        W[n].requires_grad_(n in names)
    params = [W[n] for n in names]
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: lr_lambda(s, warmup, total))
    out = []
    for batch in batches:
        loss, logits, acc, _ = slam_forward(W, cfg, batch)
        loss.backward()
        grads = {n: W[n].grad.detach().clone() for n in names}
        opt.step()
        sched.step()
        opt.zero_grad()
        out.append({"loss": loss.detach(), "acc": acc, "grads": grads})
    return out


# ---------------------------------------------------------------------------------------------- f4: AnyPrecisionAdamW
def anyprecision_adamw_step(p, grad, state: dict, lr: float, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                            use_kahan_summation=False, state_dtype=torch.bfloat16):
    """One step of AnyPrecisionAdamW (src/slam_llm/policies/anyprecision_optimizer.py:73-178) on one tensor, in place: every
    tensor op rounds to ITS tensor's dtype, in the reference's order -- decoupled weight decay on p (:128-129), momentum
    `mul_(b1).add_(g, alpha=1-b1)` (:132, two roundings), variance `mul_(b2).addcmul_(g, g, value=1-b2)` (:135),
    `(sqrt(v) / sqrt(1 - b2^t)).add_(eps)` (:147-149, three roundings), then `p.addcdiv_(m, denom, value=-lr/(1-b1^t))` (:165) or
    the Kahan form (:152-160).  `step` is a float32 0-dim TENSOR there, so 1 - b^t, the step size and the denominator correction
    are float32 values (not Python doubles).  `state` holds step / exp_avg / exp_avg_sq (/ compensation)."""
    b1, b2 = betas
    if not state:
        state["step"] = torch.tensor(0.0)   # a float32 0-dim tensor in the reference (:112): the bias corrections below are fp32
        state["exp_avg"] = torch.zeros_like(p, dtype=state_dtype)
        state["exp_avg_sq"] = torch.zeros_like(p, dtype=state_dtype)
        if use_kahan_summation:
            state["compensation"] = torch.zeros_like(p, dtype=state_dtype)
    state["step"] += 1
    t = state["step"]
    m, v = state["exp_avg"], state["exp_avg_sq"]
    if weight_decay:
        p.mul_(1 - lr * weight_decay)
    m.mul_(b1).add_(grad, alpha=1 - b1)
    v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
    step_size = lr / (1 - b1 ** t)
    denom = (v.sqrt() / ((1 - b2 ** t) ** 0.5)).add_(eps, alpha=1)
    if use_kahan_summation:
        c = state["compensation"]
        c.addcdiv_(m, denom, value=-step_size)
        tmp = p.detach().clone()
        p.add_(c)
        c.add_(tmp.sub_(p))
    else:
        p.addcdiv_(m, denom, value=-step_size)
    return p


# ---------------------------------------------------------------------------------------------- f1: generate (beam / greedy)
def _repetition_penalty(scores: torch.Tensor, history: torch.Tensor, penalty: float) -> torch.Tensor:
    """HF RepetitionPenaltyLogitsProcessor: every token id already in the row's history gets score*p if score < 0 else
    score/p (applied before the min-length mask; the history of an inputs_embeds prompt is the generated tokens only)."""
    if penalty == 1.0 or history.shape[1] == 0:
        return scores
    sc = torch.gather(scores, 1, history)
    sc = torch.where(sc < 0, sc * penalty, sc / penalty)
    return scores.scatter(1, history, sc)


def warp_scores(scores: torch.Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0) -> torch.Tensor:
    """HF TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper (transformers/generation/logits_process.py), the
    order `_get_logits_processor` appends them in when do_sample=True; min_tokens_to_keep = 1, filter value -inf."""
    if temperature != 1.0:
        scores = scores / temperature
    if top_k:
        kth = torch.topk(scores, min(int(top_k), scores.shape[-1]))[0][..., -1, None]
        scores = scores.masked_fill(scores < kth, -float("inf"))
    if top_p < 1.0:
        srt, idx = torch.sort(scores, descending=False)
        remove = srt.softmax(dim=-1).cumsum(dim=-1) <= (1 - top_p)
        remove[..., -1:] = False
        scores = scores.masked_fill(remove.scatter(1, idx, remove), -float("inf"))
    return scores


def greedy_search(step_fn, batch_size: int, max_new_tokens: int, eos: int, pad: int, min_length: int = 1,
                  repetition_penalty: float = 1.0, trace: Optional[list] = None, sample: Optional[dict] = None):
    """HF `GenerationMixin._sample` with do_sample=False (transformers/generation/utils.py), the num_beams=1 branch of
    `self.llm.generate(...)` at slam_model.py:438-452.  The prompt is inputs_embeds only, so the token history starts
    empty: MinLengthLogitsProcessor(min_length) therefore masks eos while fewer than `min_length` tokens exist.
    step_fn(tokens [R, t] int64, src_rows [R]) -> next-token logits [R, V] fp32."""
    toks = torch.zeros((batch_size, 0), dtype=torch.int64)
    alive = torch.ones(batch_size, dtype=torch.bool)
    rows = torch.arange(batch_size)
    while True:
        logits = _repetition_penalty(step_fn(toks, rows).float().clone(), toks, repetition_penalty)
        if toks.shape[1] < min_length:
            logits[:, eos] = -float("inf")
        if sample is not None:   # do_sample=True: warpers, softmax, one multinomial draw per row (HF `_sample`)
            nxt = torch.multinomial(F.softmax(warp_scores(logits, **sample), dim=-1), num_samples=1).squeeze(1)
        else:
            nxt = logits.argmax(-1)
        if trace is not None:   # decision margins for margin-aware comparisons of a reduced-precision path (tests)
            top2 = torch.topk(logits, 2, dim=-1)[0]
            trace.append({"margin": torch.where(alive, top2[:, 0] - top2[:, 1], torch.full((batch_size,), float("inf"))),
                          "scale": logits.abs().amax(-1) if not torch.isinf(logits).any() else logits.masked_fill(torch.isinf(logits), 0).abs().amax(-1)})
        nxt = torch.where(alive, nxt, torch.full_like(nxt, pad))
        toks = torch.cat([toks, nxt[:, None]], dim=1)
        alive = alive & (nxt != eos) & (toks.shape[1] < max_new_tokens)
        if not bool(alive.any()):
            return toks


def beam_search(step_fn, batch_size: int, num_beams: int, max_new_tokens: int, eos: int, pad: int,
                min_length: int = 1, length_penalty: float = 1.0, repetition_penalty: float = 1.0,
                trace: Optional[list] = None, sample: Optional[dict] = None):
    """HF `GenerationMixin._beam_search` (transformers 5.x vectorised form; early_stopping=False, one eos id,
    num_return_sequences=1), restated per batch item.  Each item keeps `num_beams` running hypotheses and
    `num_beams` finished ones; every step the best 2*num_beams continuations are ranked, the non-terminated ones
    refill the running set, and a terminated one enters the finished set only if it ranks inside the top
    `num_beams` continuations.  Finished scores are sum-logprob / length**length_penalty.  An item is closed once its
    best running score / cur_len**length_penalty cannot beat its worst finished score; the loop ends when every item
    is closed or max_new_tokens is reached.  Returns [batch, max generated length] padded with `pad or eos`.
    step_fn(tokens [R, t] int64, src_rows [R] = row of the previous call each hypothesis extends) -> logits [R, V]."""
    nb, L, K = num_beams, max_new_tokens, 2 * num_beams
    NEG = -1.0e9
    pad = pad or eos  # HF: `output_fill_value = pad_token_id or eos_token_id[0]` -- pad id 0 falls through to eos
    run_seq = [torch.full((nb, L), pad, dtype=torch.int64) for _ in range(batch_size)]
    run_score = [torch.tensor([0.0] + [NEG] * (nb - 1)) for _ in range(batch_size)]
    fin_seq = [torch.full((nb, L), pad, dtype=torch.int64) for _ in range(batch_size)]
    fin_score = [torch.full((nb,), NEG) for _ in range(batch_size)]
    fin_flag = [torch.zeros(nb, dtype=torch.bool) for _ in range(batch_size)]
    fin_len = [torch.zeros(nb, dtype=torch.int64) for _ in range(batch_size)]
    open_ = [True] * batch_size
    src_rows = torch.arange(batch_size * nb)
    t = 0
    while True:
        flat = torch.cat([r[:, :t] for r in run_seq], dim=0)
        logits = step_fn(flat, src_rows).float()
        lp_all = _repetition_penalty(F.log_softmax(logits, dim=-1), flat, repetition_penalty)
        if t < min_length:
            lp_all[:, eos] = -float("inf")
        V = lp_all.shape[-1]
        all_hit = True
        new_src = []
        if sample is not None:
            # do_sample=True (`_get_top_k_continuations`): the warpers run on the per-hypothesis log-probs, then K continuations
            # are DRAWN without replacement from softmax(accumulated scores) -- one batched multinomial call, order as drawn
            lp_all = warp_scores(lp_all, **sample)
            acc_all = (lp_all.view(batch_size, nb, V) + torch.stack(run_score)[:, :, None]).view(batch_size, nb * V)
            drawn = torch.multinomial(F.softmax(acc_all, dim=-1), num_samples=K)
        for b in range(batch_size):
            acc = (lp_all[b * nb:(b + 1) * nb] + run_score[b][:, None]).reshape(-1)
            if sample is not None:
                top_idx = drawn[b]
                top_lp = acc[top_idx]
            else:
                top_lp, top_idx = torch.topk(acc, K)
            if trace is not None and open_[b]:
                # every ranking decision of the step is a comparison between neighbours of the sorted top K+1 candidates
                srt = torch.topk(acc, K + 1)[0]
                live = srt > NEG / 2
                gaps = (srt[:-1] - srt[1:])[live[:-1] & live[1:]]
                trace.append({"item": b, "t": t, "margin": float(gaps.min()) if gaps.numel() else float("inf"),
                              "scale": float(logits[b * nb:(b + 1) * nb].abs().max())})
            src, tok = top_idx // V, top_idx % V
            cand = run_seq[b][src].clone()
            cand[:, t] = tok
            hits = (tok == eos) | (t + 1 >= L)
            all_hit = all_hit and bool(hits.all())
            # running set: best non-terminated continuations
            run_lp = top_lp + hits.float() * NEG
            sel = torch.topk(run_lp, nb)[1]
            run_seq[b], run_score[b] = cand[sel], run_lp[sel]
            new_src.append(src[sel] + b * nb)
            # finished set
            just = hits & (torch.arange(K) < nb)
            sc = top_lp / ((t + 1) ** length_penalty)
            sc = sc + (0.0 if open_[b] else 1.0) * NEG
            sc = sc + (~just).float() * NEG
            m_score = torch.cat([fin_score[b], sc])
            idx = torch.topk(m_score, nb)[1]
            fin_seq[b] = torch.cat([fin_seq[b], cand])[idx]
            fin_score[b] = m_score[idx]
            fin_flag[b] = torch.cat([fin_flag[b], just])[idx]
            fin_len[b] = torch.cat([fin_len[b], torch.full((K,), t + 1, dtype=torch.int64)])[idx]
        src_rows = torch.cat(new_src)
        t += 1
        for b in range(batch_size):
            best = run_score[b][:1] / (t ** length_penalty)
            worst = torch.where(fin_flag[b], fin_score[b].min(), torch.tensor(NEG))
            open_[b] = open_[b] and bool((best > worst).any())
        if not any(open_) or all_hit:
            break
    out_len = int(max(int(fl[0]) for fl in fin_len))
    return torch.stack([fs[0, :out_len] for fs in fin_seq])


def generate_position_ids(attention_mask: torch.Tensor) -> torch.Tensor:
    """HF prepare_inputs_for_generation: position_ids = cumsum(attention_mask) - 1, pad slots set to 1."""
    pos = attention_mask.long().cumsum(-1) - 1
    return pos.masked_fill(attention_mask == 0, 1)


def slam_generate(W, cfg, batch: dict, max_new_tokens=200, num_beams=4, min_length=1, length_penalty=1.0,
                  eos=2, pad=0, repetition_penalty=1.0, trace: Optional[list] = None, sample: Optional[dict] = None):
    """slam_model.generate (src/slam_llm/models/slam_model.py:409-456): forward(..., inference_mode=True) returns
    (inputs_embeds, attention_mask) [slam_model.py:394-395], then `self.llm.generate(inputs_embeds=...,
    attention_mask=..., num_beams, max_new_tokens, min_length, length_penalty, eos/pad ids)`.  sample=None is
    do_sample=False; sample={temperature, top_k, top_p} is do_sample=True (draws from torch's global CPU generator exactly as HF
    does: seed it first).  repetition_penalty is HF's RepetitionPenaltyLogitsProcessor.  The oracle re-runs the full sequence every step
    (no KV cache) in fp32."""
    mel = batch["audio_mel"]
    enc = whisper_encoder(W, cfg, mel.permute(0, 2, 1))
    proj = projector_concat(W, enc, cfg["ds_rate"])
    emb_w = W["llm.base_model.model.model.embed_tokens.weight"]
    embeds = embed_splice(emb_w, batch["input_ids"].clone(), batch["modality_mask"].bool(), proj)
    mask = batch["attention_mask"].long()
    B = embeds.shape[0]

    def step_fn(tokens, src_rows):
        # row r of this call extends row src_rows[r] of the previous call; prompts only depend on the batch item
        R = tokens.shape[0]
        item = torch.arange(R) // (R // B)
        x = torch.cat([embeds[item], F.embedding(tokens, emb_w)], dim=1)
        m = torch.cat([mask[item], torch.ones_like(tokens)], dim=1)
        _, logits = llama_forward(W, cfg, x, m, None, position_ids=generate_position_ids(m))
        return logits[:, -1, :]

    if num_beams == 1:
        return greedy_search(step_fn, B, max_new_tokens, eos, pad, min_length, repetition_penalty, trace, sample)
    return beam_search(step_fn, B, num_beams, max_new_tokens, eos, pad, min_length, length_penalty, repetition_penalty, trace,
                       sample)


# ---------------------------------------------------------------------------------------------- a9: batcher + collators
def window_class(elem_len: int, buffer_lens: List[int], max_frame_length: int) -> bool:
    """src/slam_llm/datasets/speech_dataset_large.py:259-263 on sequence lengths."""
    if len(buffer_lens) == 0:
        return True
    return (len(buffer_lens) + 1) * max(elem_len, max(buffer_lens)) > max_frame_length


def dynamic_batches(lengths: List[int], max_frame_length: int) -> List[List[int]]:
    """MultiTaskDynamicBatchDataset.__iter__, speech_dataset_large.py:244-256: returns index groups."""
    out, buf = [], []
    for i, n in enumerate(lengths):
        if not window_class(n, [lengths[j] for j in buf], max_frame_length):
            buf.append(i)
        else:
            if buf:
                out.append(buf)
            buf = [i]
    if buf:
        out.append(buf)
    return out


def make_sample(audio_length: int, prompt_ids: List[int], answer_ids: List[int], eos: int):
    """token layout of SpeechDatasetJsonl.__getitem__, src/slam_llm/datasets/speech_dataset.py:109-161."""
    ids = torch.tensor([-1] * audio_length + list(prompt_ids) + list(answer_ids) + [eos], dtype=torch.int64)
    labels = ids.clone()
    labels[: audio_length + len(prompt_ids)] = -1
    mask = ids.ge(-1)
    label_mask = labels.ge(0)
    ids[~mask] = 0
    labels[~label_mask] = -100
    return {"input_ids": ids, "labels": labels, "attention_mask": mask, "audio_length": audio_length,
            "prompt_length": len(prompt_ids)}


def collate_left_pad(samples: List[dict], pad_id: int, mels: Optional[List[torch.Tensor]] = None) -> dict:
    """SpeechDatasetJsonl.collator, speech_dataset.py:216-291: [audio,prompt] left-padded, answer right-padded."""
    pl = [s["audio_length"] + s["prompt_length"] for s in samples]
    al = [len(s["input_ids"]) - p for s, p in zip(samples, pl)]
    pm, am = max(pl), max(al)

    def pad(t, left, right, v):
        return torch.cat([torch.full((left,), v, dtype=t.dtype), t, torch.full((right,), v, dtype=t.dtype)])

    out = {
        "input_ids": torch.stack([pad(s["input_ids"], pm - p, am - a, pad_id) for s, p, a in zip(samples, pl, al)]),
        "labels": torch.stack([pad(s["labels"], pm - p, am - a, -100) for s, p, a in zip(samples, pl, al)]),
        "attention_mask": torch.stack([pad(s["attention_mask"], pm - p, am - a, False) for s, p, a in zip(samples, pl, al)]),
    }
    mm = torch.zeros_like(out["attention_mask"])
    for i, (s, p) in enumerate(zip(samples, pl)):
        mm[i, pm - p: pm - p + s["audio_length"]] = True
    out["modality_mask"] = mm
    if mels is not None:
        tm = max(m.shape[0] for m in mels)
        out["audio_mel"] = torch.stack([F.pad(m, (0, 0, 0, tm - m.shape[0])) for m in mels])
        pm_ = torch.zeros(len(mels), (tm + 1) // 2)
        for i, m in enumerate(mels):
            pm_[i, : (m.shape[0] + 1) // 2] = 1
        out["audio_mel_post_mask"] = pm_
    return out


def collate_right_pad(samples: List[dict], pad_id: int, mels: Optional[List[torch.Tensor]] = None) -> dict:
    """MultiTaskDataset.collator, speech_dataset_large.py:180-233: right padding only, audio at [0, audio_length)."""
    tm = max(len(s["input_ids"]) for s in samples)

    def pad(t, v):
        return torch.cat([t, torch.full((tm - len(t),), v, dtype=t.dtype)])

    out = {
        "input_ids": torch.stack([pad(s["input_ids"], pad_id) for s in samples]),
        "labels": torch.stack([pad(s["labels"], -100) for s in samples]),
        "attention_mask": torch.stack([pad(s["attention_mask"], False) for s in samples]),
    }
    mm = torch.zeros_like(out["attention_mask"])
    for i, s in enumerate(samples):
        mm[i, : s["audio_length"]] = True
    out["modality_mask"] = mm
    if mels is not None:
        t2 = max(m.shape[0] for m in mels)
        out["audio_mel"] = torch.stack([F.pad(m, (0, 0, 0, t2 - m.shape[0])) for m in mels])
    return out


# ---------------------------------------------------------------------------------------------- synthetic weights / batches
def make_config(**kw) -> dict:
    cfg = dict(n_mels=80, enc_dim=128, enc_heads=2, enc_layers=2, enc_ctx=1500, ds_rate=5, proj_hidden=2048,
               llm_dim=128, llm_layers=2, llm_heads=2, llm_kv_heads=1, llm_head_dim=64, llm_ffn=256, vocab=512,
               rope_theta=10000.0, rms_eps=1e-5, lora_r=8, lora_alpha=32, lora_targets=("q_proj", "v_proj"))
    cfg.update(kw)
    return cfg


def init_weights(cfg: dict, seed: int = 42, lora_b_std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Seeded random weights at the reference's state_dict names (LoRA B non-zero so adapters contribute)."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    W = {}
    d, nm = cfg["enc_dim"], cfg["n_mels"]
    W["encoder.conv1.weight"] = rn(d, nm, 3, std=(1.0 / (3 * nm)) ** 0.5)
    W["encoder.conv1.bias"] = rn(d)
    W["encoder.conv2.weight"] = rn(d, d, 3, std=(1.0 / (3 * d)) ** 0.5)
    W["encoder.conv2.bias"] = rn(d)
    W["encoder.positional_embedding"] = sinusoids(cfg["enc_ctx"], d)
    for i in range(cfg["enc_layers"]):
        p = f"encoder.blocks.{i}."
        s = d ** -0.5
        for nme in ("query", "key", "value", "out"):
            W[p + f"attn.{nme}.weight"] = rn(d, d, std=s)
            if nme != "key":
                W[p + f"attn.{nme}.bias"] = rn(d)
        W[p + "attn_ln.weight"] = 1 + rn(d, std=0.1)
        W[p + "attn_ln.bias"] = rn(d, std=0.1)
        W[p + "mlp.0.weight"] = rn(4 * d, d, std=s)
        W[p + "mlp.0.bias"] = rn(4 * d)
        W[p + "mlp.2.weight"] = rn(d, 4 * d, std=(4 * d) ** -0.5)
        W[p + "mlp.2.bias"] = rn(d)
        W[p + "mlp_ln.weight"] = 1 + rn(d, std=0.1)
        W[p + "mlp_ln.bias"] = rn(d, std=0.1)
    W["encoder.ln_post.weight"] = 1 + rn(d, std=0.1)
    W["encoder.ln_post.bias"] = rn(d, std=0.1)
    k, ph, dl = cfg["ds_rate"], cfg["proj_hidden"], cfg["llm_dim"]
    W["encoder_projector.linear1.weight"] = rn(ph, d * k, std=(d * k) ** -0.5)
    W["encoder_projector.linear1.bias"] = rn(ph)
    W["encoder_projector.linear2.weight"] = rn(dl, ph, std=ph ** -0.5)
    W["encoder_projector.linear2.bias"] = rn(dl)
    Hq, Hkv, D, Fd, V = cfg["llm_heads"], cfg["llm_kv_heads"], cfg["llm_head_dim"], cfg["llm_ffn"], cfg["vocab"]
    P = "llm.base_model.model."
    W[P + "model.embed_tokens.weight"] = rn(V, dl, std=1.0)
    shapes = {"q_proj": (Hq * D, dl), "k_proj": (Hkv * D, dl), "v_proj": (Hkv * D, dl), "o_proj": (dl, Hq * D),
              "gate_proj": (Fd, dl), "up_proj": (Fd, dl), "down_proj": (dl, Fd)}
    for i in range(cfg["llm_layers"]):
        p = f"{P}model.layers.{i}."
        for nme, (o, ii) in shapes.items():
            mod = "self_attn." if nme in ("q_proj", "k_proj", "v_proj", "o_proj") else "mlp."
            W[p + mod + nme + ".weight"] = rn(o, ii, std=ii ** -0.5)
            if nme in cfg["lora_targets"]:
                W[p + mod + nme + ".lora_A.default.weight"] = rn(cfg["lora_r"], ii, std=ii ** -0.5)
                W[p + mod + nme + ".lora_B.default.weight"] = rn(o, cfg["lora_r"], std=lora_b_std)
        W[p + "input_layernorm.weight"] = 1 + rn(dl, std=0.1)
        W[p + "post_attention_layernorm.weight"] = 1 + rn(dl, std=0.1)
    W[P + "model.norm.weight"] = 1 + rn(dl, std=0.1)
    W[P + "lm_head.weight"] = rn(V, dl, std=dl ** -0.5)
    return W


def synth_audio(n_clips: int, seconds: float, seed: int = 1234) -> torch.Tensor:
    """SURVEY 8d synthetic audio: N(0, 0.1^2) clamped to [-1, 1], fp32, 16 kHz."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(n_clips, int(seconds * SAMPLE_RATE), generator=g) * 0.1).clamp(-1, 1)


def synth_batch(cfg, audio: torch.Tensor, prompt_len=16, answer_lens=(64,), seed=1236, left_pad=True,
                pad_to_30s=True) -> dict:
    """One synthetic batch in the reference's dict layout."""
    g = torch.Generator().manual_seed(seed)
    V = cfg["vocab"]
    samples, mels = [], []
    for i in range(audio.shape[0]):
        a = pad_or_trim(audio[i]) if pad_to_30s else audio[i][: audio.shape[1] // HOP * HOP]
        mel = log_mel_spectrogram(a, cfg["n_mels"]).permute(1, 0)
        mels.append(mel)
        alen = ((mel.shape[0] + 1) // 2) // cfg["ds_rate"]
        al = answer_lens[i % len(answer_lens)]
        pids = torch.randint(3, V, (prompt_len,), generator=g).tolist()
        aids = torch.randint(3, V, (al - 1,), generator=g).tolist()
        samples.append(make_sample(alen, pids, aids, eos=2))
    coll = collate_left_pad if left_pad else collate_right_pad
    return coll(samples, pad_id=2, mels=mels)


def synth_infer_batch(cfg, audio: torch.Tensor, clip_samples=(32000,), prompt_lens=(6,), seed=1237) -> dict:
    """Inference-mode batch (speech_dataset.py:120-134 + collator :259-273): [audio, prompt] only, left padded,
    no labels.  Row i uses the first clip_samples[i] samples of audio[i] (ragged clips -> ragged audio_length)."""
    g = torch.Generator().manual_seed(seed)
    samples, mels = [], []
    for i in range(audio.shape[0]):
        n = clip_samples[i % len(clip_samples)] // HOP * HOP
        mel = log_mel_spectrogram(audio[i][:n], cfg["n_mels"]).permute(1, 0)
        mels.append(mel)
        alen = ((mel.shape[0] + 1) // 2) // cfg["ds_rate"]
        pids = torch.randint(3, cfg["vocab"], (prompt_lens[i % len(prompt_lens)],), generator=g)
        ids = torch.cat([torch.zeros(alen, dtype=torch.int64), pids])
        samples.append({"input_ids": ids, "labels": torch.full_like(ids, -100), "attention_mask": torch.ones_like(ids).bool(),
                        "audio_length": alen, "prompt_length": len(pids)})
    out = collate_left_pad(samples, pad_id=2, mels=mels)
    del out["labels"]
    return out
