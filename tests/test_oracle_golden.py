"""CPU: pins oracle/slam_oracle.py against fixtures produced by the reference itself (oracle/make_golden.py).

fp32 vs fp32, different op order only -> tight tolerances (stated per check)."""
import numpy as np
import pytest
import torch

from oracle import slam_oracle as O
from oracle.make_golden_cases import CASES
from tests import golden_util as G


def test_logmel_matches_reference_feature_extractor():
    fx = G.load("logmel")
    audio = torch.from_numpy(fx["audio"])
    for nm in (80, 128):
        mel = torch.stack([O.log_mel_spectrogram(O.pad_or_trim(a), nm) for a in audio]).numpy()  # [2, nm, 3000]
        idx = fx[f"mel{nm}_frames"]
        # tolerance: SURVEY 8c "mel fp32 abs <= 1e-4"
        np.testing.assert_allclose(mel[:, :, idx], fx[f"mel{nm}_values"], atol=1e-4, rtol=0)
        np.testing.assert_allclose(mel.reshape(2, -1).max(axis=1), fx[f"mel{nm}_max"], atol=1e-5)


@pytest.mark.parametrize("name", list(CASES))
def test_forward_backward_and_optimizer_match_reference(name):
    fx = G.load(name)
    cfg = CASES[name]["cfg"]
    W = O.init_weights(cfg, seed=42)
    batch = {k[len("batch."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("batch.")}
    # the batch itself is rebuilt by the oracle's collator restatement and must equal the stored one
    audio = torch.from_numpy(fx["audio"])
    rb = O.synth_batch(cfg, audio, prompt_len=6, answer_lens=CASES[name]["answer_lens"], seed=1236,
                       left_pad=CASES[name]["left_pad"], pad_to_30s=False)
    for k in ("input_ids", "labels", "attention_mask", "modality_mask"):
        assert torch.equal(rb[k], batch[k]), k
    outs = O.train_steps(W, cfg, [dict(batch) for _ in range(3)], lr=1e-2, weight_decay=0.01, warmup=2, total=10)
    for s in range(3):
        assert abs(float(outs[s]["loss"]) - float(fx[f"loss.{s}"])) < 2e-5, (s, float(outs[s]["loss"]), float(fx[f"loss.{s}"]))
        assert abs(float(outs[s]["acc"]) - float(fx[f"acc.{s}"])) < 1e-6
    # first step at lr = 0 leaves the parameters untouched: loss.0 == loss.1
    assert float(fx["loss.0"]) == float(fx["loss.1"])
    W0 = O.init_weights(cfg, seed=42)
    with torch.no_grad():
        loss, logits, acc, aux = O.slam_forward(W0, cfg, dict(batch))
    G.check_packed(fx, "encoder_out", aux["encoder_out"].numpy(), atol=2e-5, rtol=1e-4)
    G.check_packed(fx, "projector_out", aux["projector_out"].numpy(), atol=2e-5, rtol=1e-4)
    # logits of pad query rows are garbage-by-design in the reference but deterministic in fp32
    G.check_packed(fx, "logits", logits.numpy(), atol=5e-5, rtol=1e-4)
    for n, g in outs[0]["grads"].items():
        G.check_packed(fx, "grad." + n, g.numpy(), atol=1e-6, rtol=1e-3, norm_rtol=1e-4)
    for n in O.trainable_names(W):
        # Adam normalises the update (lr * m / sqrt(v)): elements with a near-zero gradient move by O(lr) on
        # rounding-level gradient differences, so elementwise bounds are a fraction of the total movement
        # (2 real steps x lr 1e-2) and the direction of the whole update is checked by cosine.
        gold, mine = G.sub(fx, "final." + n, W[n].detach().numpy())
        _, init = G.sub(fx, "final." + n, W0[n].detach().numpy())
        assert np.abs(gold - mine).max() < 2e-3, n
        if np.abs(gold - init).max() > 0:
            assert G.cosine(gold - init, mine - init) > 0.999, n


def test_unfrozen_encoder_training_matches_reference():
    """train_config.freeze_encoder=false: the oracle's autograd through its Whisper restatement == the reference's slam_model
    with a trainable encoder (fixture written by oracle/make_golden_unfrozen.py): losses, every gradient, final parameters"""
    from oracle.make_golden_cases import UNFROZEN_CASE as C
    fx = G.load("step_unfrozen")
    cfg = C["cfg"]
    W = O.init_weights(cfg, seed=42)
    batch = {k[len("batch."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("batch.")}
    outs = O.train_steps(W, cfg, [dict(batch) for _ in range(3)], lr=C["lr"], weight_decay=0.01, warmup=2, total=10, train_encoder=True)
    for s in range(3):
        assert abs(float(outs[s]["loss"]) - float(fx[f"loss.{s}"])) < 2e-5
    enc_names = [n for n in outs[0]["grads"] if n.startswith("encoder.")]
    assert len(enc_names) == 4 + 15 * cfg["enc_layers"] + 2 and "encoder.positional_embedding" not in outs[0]["grads"]
    for n, g in outs[0]["grads"].items():
        # key-projection rows feed a softmax that is invariant to per-query constants: their bias does not exist, fine
        G.check_packed(fx, "grad." + n, g.numpy(), atol=1e-6, rtol=1e-3, norm_rtol=1e-4)
    W0 = O.init_weights(cfg, seed=42)
    for n in outs[0]["grads"]:
        gold, mine = G.sub(fx, "final." + n, W[n].detach().numpy())
        _, init = G.sub(fx, "final." + n, W0[n].detach().numpy())
        assert np.abs(gold - mine).max() < 2e-3, n
        if np.abs(gold - init).max() > 0:
            assert G.cosine(gold - init, mine - init) > 0.999, n


def test_dynamic_batcher_matches_reference_window_class():
    # fixtures come from the reference's own MultiTaskDynamicBatchDataset/window_class (speech_dataset_large.py:235-263)
    fx = G.load("batcher")
    ci = 0
    while f"lens.{ci}" in fx.files:
        groups = O.dynamic_batches([int(x) for x in fx[f"lens.{ci}"]], int(fx[f"mfl.{ci}"]))
        assert [len(g) for g in groups] == [int(x) for x in fx[f"group_sizes.{ci}"]], ci
        assert sum(groups, []) == list(range(len(fx[f"lens.{ci}"])))
        ci += 1
    assert ci == 4
    # 31 samples of T=380 fit under 12000 and the 32nd does not
    assert [int(x) for x in fx["group_sizes.1"]] == [31, 31, 2]


def test_hubert_encoder_matches_hf_twin_of_reference():
    from oracle.make_golden_cases import HUBERT_TINY
    fx = G.load("hubert_tiny")
    W = O.init_hubert_weights(HUBERT_TINY, seed=7)
    with torch.no_grad():
        out = O.hubert_encoder(W, HUBERT_TINY, torch.from_numpy(fx["wav"]))
    assert list(out.shape) == [int(x) for x in fx["out_shape"]]
    G.check_packed(fx, "out", out.numpy(), atol=3e-5, rtol=1e-4)


def test_hubert_ragged_batch_matches_masked_twin():
    """ragged raw-audio batch: oracle (fairseq frame-mask rule + zeroed padded frames + key masking) == HF HubertModel run
    with that frame mask (fixture), on the valid frames"""
    from oracle.make_golden_cases import HUBERT_TINY
    fx = G.load("hubert_tiny_ragged")
    W = O.init_hubert_weights(HUBERT_TINY, seed=7)
    nv = torch.from_numpy(fx["n_valid"])
    with torch.no_grad():
        out = O.hubert_encoder(W, HUBERT_TINY, torch.from_numpy(fx["wav"]), n_valid=nv)
    pad = O.hubert_frame_padding_mask(fx["wav"].shape[1], out.shape[1], nv)
    assert np.array_equal(pad.numpy(), fx["frame_padding_mask"]) and (~pad).sum(1).tolist() == [49, 28, 38]
    G.check_packed(fx, "out", out.masked_fill(pad[:, :, None], 0.0).numpy(), atol=3e-5, rtol=1e-4)


def test_hubert_base_encoder_matches_hf_twin():
    """HuBERT-base structure (GroupNorm over time after the first conv only, no conv bias, post-LN layers): oracle == HF HubertModel
    configured feat_extract_norm="group", do_stable_layer_norm=False (fixture: oracle/make_golden_hubert_base.py)"""
    from oracle.make_golden_cases import HUBERT_BASE_TINY as C
    fx = G.load("hubert_base_tiny")
    W = O.init_hubert_weights(C, seed=8)
    assert not any(k.endswith("conv.bias") and "feature_extractor" in k for k in W)
    with torch.no_grad():
        out = O.hubert_encoder(W, C, torch.from_numpy(fx["wav"]))
        nv = torch.from_numpy(fx["ragged.n_valid"])
        out_r = O.hubert_encoder(W, C, torch.from_numpy(fx["ragged.wav"]), n_valid=nv)
    assert list(out.shape) == [int(x) for x in fx["out_shape"]]
    G.check_packed(fx, "out", out.numpy(), atol=3e-5, rtol=1e-4)
    pad = O.hubert_frame_padding_mask(fx["ragged.wav"].shape[1], out_r.shape[1], nv)
    assert np.array_equal(pad.numpy(), fx["ragged.frame_padding_mask"])
    G.check_packed(fx, "ragged.out", out_r.masked_fill(pad[:, :, None], 0.0).numpy(), atol=3e-5, rtol=1e-4)


def test_wavlm_encoder_matches_the_reference_module():
    """oracle.wavlm_encoder == the reference's own WavLM (models/wavlm/WavLM.py, fixture written by oracle/make_golden_wavlm.py):
    equal-length and ragged zero-padded batches; the product's host-side bucket table == the oracle's bucket function"""
    from oracle.make_golden_cases import WAVLM_TINY as C
    from slam_llm_amd.host_tables import wavlm_relative_buckets
    fx = G.load("wavlm_tiny")
    W = O.init_wavlm_weights(C, seed=9)
    with torch.no_grad():
        out = O.wavlm_encoder(W, C, torch.from_numpy(fx["wav"]))
        nv = torch.from_numpy(fx["ragged.n_valid"])
        out_r = O.wavlm_encoder(W, C, torch.from_numpy(fx["ragged.wav"]), n_valid=nv)
    assert list(out.shape) == [int(x) for x in fx["out_shape"]]
    G.check_packed(fx, "out", out.numpy(), atol=3e-5, rtol=1e-4)
    pad = O.hubert_frame_padding_mask(fx["ragged.wav"].shape[1], out_r.shape[1], nv)
    assert np.array_equal(pad.numpy(), fx["ragged.frame_padding_mask"])
    G.check_packed(fx, "ragged.out", out_r.masked_fill(pad[:, :, None], 0.0).numpy(), atol=3e-5, rtol=1e-4)
    for T, nb, md in ((49, 40, 24), (1500, 320, 800), (7, 320, 800)):
        rel = torch.arange(-(T - 1), T)
        assert torch.equal(wavlm_relative_buckets(T, nb, md), O.wavlm_relative_buckets(rel, nb, md))
        assert int(wavlm_relative_buckets(T, nb, md).max()) < nb


@pytest.mark.parametrize("tag", ["A", "B", "C", "D", "E"])
def test_wavlm_train_mode_regularisers_match_the_reference_module(tag):
    """the un-frozen WavLM is left in train mode (slam_model.py:317-318): dropout_input, the dropout after the positional conv, per
    layer attention_dropout / dropout1 / dropout2 / dropout3 and layerdrop, with the masks the reference module drew handed to the
    oracle (fixture written by oracle/make_golden_wavlm_train.py from the reference's own WavLM in .train()).  A: layer 1 skipped on a
    ragged batch; B: layer 0 skipped -- no position bias is ever created, the other layers run without bias and gate; C: all kept;
    D / E: the Base structure (group-norm extractor, post-LN layers), all kept on a ragged batch / layer 1 skipped.
    Output and every parameter gradient."""
    from oracle.make_golden_cases import WAVLM_BASE_TINY, WAVLM_TRAIN_TINY
    C, wseed = (WAVLM_TRAIN_TINY, 9) if tag in "ABC" else (WAVLM_BASE_TINY, 10)
    fx = G.load("wavlm_train_tiny")
    W = {k: v.requires_grad_(True) for k, v in O.init_wavlm_weights(C, seed=wseed).items()}
    tr = G.wavlm_train_masks(fx, tag, C["hub_layers"])
    nv = torch.from_numpy(fx[tag + ".n_valid"])
    wav = torch.from_numpy(fx[tag + ".wav"])
    ragged = bool((nv != wav.shape[1]).any())
    out = O.wavlm_encoder(W, C, wav, n_valid=nv if ragged else None, train=tr)
    pad = O.hubert_frame_padding_mask(wav.shape[1], out.shape[1], nv)
    G.check_packed(fx, tag + ".out", out.detach().masked_fill(pad[:, :, None], 0.0).numpy(), atol=3e-5, rtol=1e-4)
    (out * torch.from_numpy(fx[tag + ".cot"])).sum().backward()
    gmax = max(float(fx[f"{tag}.grad.{k}.__norm"]) for k in W if f"{tag}.grad.{k}.__norm" in fx)
    n_none = 0
    for k in W:
        if f"{tag}.grad.{k}.__none" in fx:          # parameters of the skipped layer, mask_emb; in B also every gate and the bucket table
            assert W[k].grad is None or float(W[k].grad.abs().max()) == 0.0, k
            n_none += 1
            continue
        gn = float(fx[f"{tag}.grad.{k}.__norm"])
        if k.endswith("k_proj.bias") and gn < 1e-5 * gmax:
            continue
        G.check_packed(fx, f"{tag}.grad.{k}", W[k].grad.numpy(), atol=1e-6 * gmax, rtol=2e-3)
    kept = [bool(x) for x in fx[tag + ".kept"]]
    assert n_none >= 1 + 19 * kept.count(False) and (tag != "B" or n_none >= 1 + 19 + 1 + 3 * 2)
    assert len(kept) == C["hub_layers"]


def test_wavlm_base_encoder_matches_the_reference_module():
    """the Base / Base+ structure (extractor_mode "default": GroupNorm over time after the first conv only; post-LN layers with the
    encoder-level LayerNorm in front of them): oracle == the reference's own WavLM on the fixture it wrote, equal-length and ragged"""
    from oracle.make_golden_cases import WAVLM_BASE_TINY as C
    assert C["hub_extractor_mode"] == "default" and C["hub_layer_norm_first"] is False
    fx = G.load("wavlm_base_tiny")
    W = O.init_wavlm_weights(C, seed=10)
    assert not any(k.endswith("2.1.weight") for k in W) and sum(k.endswith("conv_layers.0.2.weight") for k in W) == 1
    with torch.no_grad():
        out = O.wavlm_encoder(W, C, torch.from_numpy(fx["wav"]))
        nv = torch.from_numpy(fx["ragged.n_valid"])
        out_r = O.wavlm_encoder(W, C, torch.from_numpy(fx["ragged.wav"]), n_valid=nv)
    assert list(out.shape) == [int(x) for x in fx["out_shape"]]
    G.check_packed(fx, "out", out.numpy(), atol=3e-5, rtol=1e-4)
    pad = O.hubert_frame_padding_mask(fx["ragged.wav"].shape[1], out_r.shape[1], nv)
    assert np.array_equal(pad.numpy(), fx["ragged.frame_padding_mask"])
    G.check_packed(fx, "ragged.out", out_r.masked_fill(pad[:, :, None], 0.0).numpy(), atol=3e-5, rtol=1e-4)


def test_qformer_projector_matches_reference_module():
    from oracle.make_golden_cases import QFORMER_CASE as C
    fx = G.load("qformer")
    W = O.init_qformer_weights(C["cfg"], C["enc_dim"], C["llm_dim"], seed=11)
    for v in W.values():
        v.requires_grad_(True)
    out = O.projector_qformer(W, C["cfg"], torch.from_numpy(fx["x"]), torch.from_numpy(fx["atts"]))
    G.check_packed(fx, "out", out.detach().numpy(), atol=3e-5, rtol=1e-4)
    (out * torch.from_numpy(fx["cot"])).sum().backward()
    for n, v in W.items():
        # key biases have a mathematically zero gradient (softmax is invariant to a per-query constant): pure
        # rounding noise, compared in absolute terms only
        tiny = float(fx["grad." + n + ".__norm"]) < 1e-4
        G.check_packed(fx, "grad." + n, v.grad.numpy(), atol=2e-5, rtol=2e-3, norm_rtol=None if tiny else 1e-3)


def test_cov1d_projector_matches_reference_module():
    from oracle.make_golden_cases import COV1D_CASE as C
    fx = G.load("cov1d")
    W = O.init_cov1d_weights(C["enc_dim"], C["llm_dim"], C["k"])
    for v in W.values():
        v.requires_grad_(True)
    out = O.projector_cov1d(W, torch.from_numpy(fx["x"]), C["k"])
    G.check_packed(fx, "out", out.detach().numpy(), atol=2e-5, rtol=1e-4)
    (out * torch.from_numpy(fx["cot"])).sum().backward()
    for n, v in W.items():
        G.check_packed(fx, "grad." + n, v.grad.numpy(), atol=2e-5, rtol=1e-3, norm_rtol=1e-4)


GEN_RUNS = ((1, 1.0, 0, 1.0), (4, 1.0, 0, 1.0), (4, 2.0, 1, 1.0), (3, 0.0, 1, 1.0), (1, 1.0, 1, 1.3), (4, 1.0, 1, 1.3))


def gen_key(scale, nb, lp, pad, rp):
    return f"s{scale}.tokens.b{nb}.lp{lp}.pad{pad}" + (f".rp{rp}" if rp != 1.0 else "")


def generate_case_weights(scale):
    from oracle.make_golden_cases import GENERATE_CASE
    W = O.init_weights(GENERATE_CASE["cfg"], seed=42)
    W["llm.base_model.model.lm_head.weight"] = W["llm.base_model.model.lm_head.weight"] * scale
    return W


@pytest.mark.parametrize("scale", [24.0, 5.0])
def test_generate_matches_reference_generate(scale):
    """oracle greedy/beam restatement == slam_model.generate -> HF generate (fixture written by the reference)"""
    from oracle.make_golden_cases import GENERATE_CASE as C
    fx = G.load("generate")
    W = generate_case_weights(scale)
    batch = {k[len("batch."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("batch.")}
    eos = int(fx[f"s{scale}.eos"])
    for nb, lp, pad, rp in GEN_RUNS:
        got = O.slam_generate(W, C["cfg"], {k: v.clone() for k, v in batch.items()}, max_new_tokens=C["max_new_tokens"],
                              num_beams=nb, length_penalty=lp, eos=eos, pad=pad, repetition_penalty=rp)
        want = fx[gen_key(scale, nb, lp, pad, rp)]
        assert got.shape == want.shape and (got.numpy() == want).all(), (nb, lp, pad, got, want)


ANYPRECISION_CASES = {"fp32_params": (torch.float32, False, 0.01), "bf16_params": (torch.bfloat16, False, 0.01),
                      "bf16_params_kahan": (torch.bfloat16, True, 0.0)}


@pytest.mark.parametrize("name", sorted(ANYPRECISION_CASES))
def test_anyprecision_adamw_restatement_matches_reference_class(name):
    """oracle.anyprecision_adamw_step == the reference's AnyPrecisionAdamW (fixture written by the reference class itself,
    oracle/make_golden_anyprecision.py): parameters and bf16 states bit for bit over 6 steps"""
    fx = G.load("anyprecision")
    pdt, kahan, wd = ANYPRECISION_CASES[name]
    p = torch.from_numpy(fx["p0"]).to(pdt).clone()
    state = {}
    for s in range(6):
        g = torch.from_numpy(fx[f"grad.{s}"]).to(pdt)
        O.anyprecision_adamw_step(p, g, state, float(fx["lr"]), weight_decay=wd, use_kahan_summation=kahan)
        assert np.array_equal(p.float().numpy(), fx[f"{name}.p.{s}"]), (name, s)
        assert np.array_equal(state["exp_avg"].float().numpy(), fx[f"{name}.m.{s}"])
        assert np.array_equal(state["exp_avg_sq"].float().numpy(), fx[f"{name}.v.{s}"])
        if kahan:
            assert np.array_equal(state["compensation"].float().numpy(), fx[f"{name}.c.{s}"])
