"""GPU: the whole HIP path (log-mel -> encoder -> projector -> splice -> LLM+LoRA -> loss -> backward -> AdamW)
against (a) fixtures produced by the reference itself and (b) the CPU oracle on the same seeded inputs.

Tolerances: bf16 path vs fp32 reference: encoder/projector rel <= 2e-2 of the tensor scale,
loss abs <= 1e-2 (bf16), logits on the stored subset within a bf16 bound, gradients cosine >= 0.999 (
every trainable tensor must pass 0.999)."""
import numpy as np
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import slam_oracle as O
from oracle.make_golden_cases import CASES
from tests import golden_util as G

pytestmark = pytest.mark.gpu


def build(cfg, dev, chunk_rows=None):
    from slam_llm_amd.model import SlamHipModel
    W = O.init_weights(cfg, seed=42)
    model = SlamHipModel(dict(cfg, lora_dropout=0.0), dev).load_weights(W)
    model.llm.lm_head_chunk_rows = chunk_rows   # rows per lm_head / CE chunk (None: one 1 GB-buffer-derived chunk at these sizes)
    return model, W


# the lm_head + CE + dh pipeline runs in row chunks (3 at the headline C3 shape); 16 rows forces >= 3 chunks (with a ragged
# last one) at fixture dims so the chunk bookkeeping (row_loss / dhN / logits slices) is pinned by the reference fixtures
CHUNKS = [None, 16]


def batch_from_fixture(fx, dev):
    b = {k[len("batch."):]: torch.from_numpy(fx[k]).to(dev) for k in fx.files if k.startswith("batch.")}
    return b


def rel_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))


@pytest.mark.parametrize("chunk", CHUNKS)
@pytest.mark.parametrize("name", list(CASES))
def test_forward_matches_reference_fixture(dev, name, chunk):
    fx = G.load(name)
    cfg = CASES[name]["cfg"]
    model, W = build(cfg, dev, chunk)
    model.train()
    b = batch_from_fixture(fx, dev)
    enc = model.encoder.forward_btc(b["audio_mel"].float().contiguous())
    g, a = G.sub(fx, "encoder_out", enc.float().cpu().numpy())
    assert rel_err(a, g) < 2e-2, f"encoder_out rel err {rel_err(a, g)}"
    model._refresh()  # bf16 compute copies of the trainable tensors
    proj = model.encoder_projector.forward_hip(enc, None)
    g, a = G.sub(fx, "projector_out", proj.float().cpu().numpy())
    assert rel_err(a, g) < 2e-2, f"projector_out rel err {rel_err(a, g)}"
    model.return_logits = True
    outputs, acc = model(**{k: v.clone() for k, v in b.items()})
    assert abs(float(outputs.loss) - float(fx["loss.0"])) < 1e-2, (float(outputs.loss), float(fx["loss.0"]))
    # logits: compare only rows whose query position is not padding (pad rows are garbage by design)
    B, T, V = [int(x) for x in fx["logits.shape"]]
    lg = outputs.logits.float().cpu().numpy()
    valid_rows = b["attention_mask"].cpu().numpy().astype(bool).reshape(-1)
    stride = int(fx["logits.__stride"])
    flat_idx = np.arange(0, B * T * V, stride)
    keep = valid_rows[flat_idx // V]
    gl, al = fx["logits"][keep], lg.reshape(-1)[::stride][keep]
    assert np.abs(gl - al).max() < 6e-2 + 2e-2 * np.abs(gl).max(), f"logits max err {np.abs(gl - al).max()}"
    assert abs(float(acc) - float(fx["acc.0"])) <= 1.0 / max(1, int((b['labels'][:, 1:] != -100).sum()))


@pytest.mark.parametrize("chunk", CHUNKS)
@pytest.mark.parametrize("name", list(CASES))
def test_gradients_and_training_steps_match_reference(dev, name, chunk):
    from slam_llm_amd.model import SlamAdamW
    fx = G.load(name)
    cfg = CASES[name]["cfg"]
    model, W = build(cfg, dev, chunk)
    if chunk:
        b0 = batch_from_fixture(fx, dev)
        assert b0["input_ids"].numel() > 2 * chunk, "fixture too small to force three lm_head chunks"
    model.train()
    b = batch_from_fixture(fx, dev)
    opt = SlamAdamW(model, lr=1e-2, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: O.lr_lambda(s, 2, 10))
    losses = []
    for step in range(3):
        outputs, acc = model(**{k: v.clone() for k, v in b.items()})
        outputs.loss.backward()
        if step == 0:
            for n, p in model.store.params.items():
                g = p.grad.float().cpu().numpy()
                gold, mine = G.sub(fx, "grad." + n, g)
                cs = G.cosine(gold, mine)
                G.floor_check(cs, 0.999, f"grad {n}: cosine {cs}")
                gn = float(fx["grad." + n + ".__norm"])
                mn = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
                assert abs(mn - gn) < 3e-2 * gn + 1e-7, f"grad {n}: norm {mn} vs {gn}"
        opt.step(); sched.step(); opt.zero_grad()
        losses.append(float(outputs.loss))
    # first step runs at lr = 0 -> loss unchanged; third loss reflects two real updates
    assert abs(losses[0] - losses[1]) < 1e-6
    for s in range(3):
        assert abs(losses[s] - float(fx[f"loss.{s}"])) < 3e-2, (s, losses[s], float(fx[f"loss.{s}"]))
    # state_dict carries exactly the trainable tensors under the reference's key names
    sd = model.state_dict()
    assert set(sd.keys()) == set(O.trainable_names(W)), set(sd.keys()) ^ set(O.trainable_names(W))


def test_unfrozen_encoder_training_matches_reference_fixture(dev):
    """f4: train_config.freeze_encoder=false -- the hand-written Whisper backward (conv stem incl. col2im, LayerNorm, attention,
    GELU MLP) against the reference's slam_model with a trainable encoder (tests/golden/step_unfrozen.npz): first-step gradient of
    every trainable tensor (encoder + projector + LoRA), three AdamW steps, and the state_dict now carries the encoder."""
    from oracle.make_golden_cases import UNFROZEN_CASE as C
    from slam_llm_amd.model import SlamAdamW, SlamHipModel
    fx = G.load("step_unfrozen")
    cfg = C["cfg"]
    W = O.init_weights(cfg, seed=42)
    model = SlamHipModel(dict(cfg, lora_dropout=0.0, freeze_encoder=False), dev).load_weights(W)
    model.train()
    b = batch_from_fixture(fx, dev)
    enc_names = [n for n in model.store.params if n.startswith("encoder.")]
    assert len(enc_names) == 4 + 15 * cfg["enc_layers"] + 2
    opt = SlamAdamW(model, lr=C["lr"], weight_decay=0.01)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: O.lr_lambda(s, 2, 10))
    losses, worst = [], 1.0
    for step in range(3):
        outputs, acc = model(**{k: v.clone() for k, v in b.items()})
        outputs.loss.backward()
        if step == 0:
            for n, p in model.store.params.items():
                g = p.grad.float().cpu().numpy()
                gold, mine = G.sub(fx, "grad." + n, g)
                cs = G.cosine(gold, mine)
                worst = min(worst, cs)
                # 0.999 like the frozen-encoder cases, except the encoder's key projections (measured 0.9988 on blocks.0): softmax is
                # invariant to a per-query constant of the scores, so dK is what is left after that part cancels -- the smallest and
                # (from bf16 dS tiles) noisiest gradient of the block
                G.floor_check(cs, (G.FLOORS["unfrozen_fixture"] if n.startswith("encoder.") else G.FLOORS["frozen"]), f"grad {n}: cosine {cs}")
                gn = float(fx["grad." + n + ".__norm"])
                mn = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
                assert abs(mn - gn) < 3e-2 * gn + 1e-7, f"grad {n}: norm {mn} vs {gn}"
        opt.step(); sched.step(); opt.zero_grad()
        losses.append(float(outputs.loss.detach()))
    print("unfrozen encoder: worst gradient cosine", worst, "losses", losses)
    assert abs(losses[0] - losses[1]) < 1e-6
    for s_ in range(3):
        assert abs(losses[s_] - float(fx[f"loss.{s_}"])) < 3e-2, (s_, losses[s_], float(fx[f"loss.{s_}"]))
    sd = model.state_dict()
    assert set(sd.keys()) == set(O.trainable_names(W)) | set(enc_names)
    # gradient accumulation through the encoder backward: a second backward adds
    out1, _ = model(**{k: v.clone() for k, v in b.items()})
    out1.loss.backward()
    g1 = model.store.grad.clone()
    out2, _ = model(**{k: v.clone() for k, v in b.items()})
    out2.loss.backward()
    assert torch.allclose(model.store.grad, 2 * g1, rtol=2e-2, atol=2e-6)
    # eval-mode forward of the trainable encoder == the frozen encoder built from the same weights, to the rounding of ONE operand: the
    # frozen encoder's query projection carries d_head^-1/2 * log2(e) (folded in fp32 at load time, round 5), so its W_q / Q are the
    # bf16 roundings of scaled values (measured 3.7e-4 on this fixture; through round 4 the two were the same kernels: < 1e-5)
    model.eval()
    frozen = SlamHipModel(dict(cfg, lora_dropout=0.0), dev).load_weights({k: v.detach().clone() for k, v in
                                                                          dict(W, **{n: p.detach().cpu() for n, p in model.store.params.items()}).items()})
    frozen.eval()
    with torch.no_grad():
        l1, _ = model(**{k: v.clone() for k, v in b.items()})
        l2, _ = frozen(**{k: v.clone() for k, v in b.items()})
    assert abs(float(l1.loss) - float(l2.loss)) < 2e-3


def test_unfrozen_encoder_rejects_unimplemented_combinations(dev):
    """freeze_encoder=false exists for the Whisper / HuBERT / WavLM graphs (round 5: also with varlen_encoder, see
    test_unfrozen_whisper_ragged_encoder_step_matches_per_clip_oracle); any other encoder name is refused loudly"""
    from oracle.make_golden_cases import UNFROZEN_CASE as C
    from slam_llm_amd.model import SlamHipModel
    with pytest.raises(NotImplementedError, match="freeze_encoder"):
        SlamHipModel(dict(C["cfg"], freeze_encoder=False, encoder_name="beats"), dev)
    SlamHipModel(dict(C["cfg"], freeze_encoder=False, varlen_encoder=True), dev)      # (constructs: no longer refused)


def test_fused_swiglu_forward_step_is_bit_identical(dev, monkeypatch):
    """SLAM_FUSED_SWIGLU_FWD=1 (gate|up product with silu(gate) * up in its epilogue + block-interleaved stash read by the backward):
    the whole step -- loss, accuracy, every gradient -- is bit-identical to the default path (product -> swiglu_fwd -> ... ->
    swiglu_bwd), at a geometry where the auto rule runs the 4-wave kernel on the gate|up product (M = 12 x 380 = 4560, N = 2 x 1024,
    K = 4096)."""
    from slam_llm_amd import model as model_mod, ops
    cfg = dict(O.make_config(), enc_layers=1, llm_dim=4096, llm_layers=1, llm_heads=32, llm_kv_heads=8, llm_head_dim=128, llm_ffn=1024,
               vocab=2048, lora_dropout=0.0)
    W = O.init_weights(cfg, seed=42)
    audio = O.synth_audio(12, 30.0, seed=77)
    ob = O.synth_batch(cfg, audio, prompt_len=16, answer_lens=(64,), seed=78, left_pad=False, pad_to_30s=True)
    gb = {k: v.to(dev) for k, v in ob.items()}
    M = ob["input_ids"].numel()
    assert M == 4560 and ops.gemm_swiglu_supported(M, 2048, 4096, 4096, 4096)
    res = []
    for fused in (False, True):
        monkeypatch.setattr(model_mod, "FUSE_SWIGLU_FWD", fused)
        m = model_mod.SlamHipModel(dict(cfg), dev).load_weights(W)
        m.train()
        assert (m.llm.layers[0].gu.Wil is not None) == fused
        ops.TIMER = ops.KernelTimer()
        try:
            out, acc = m(**{k: v.clone() for k, v in gb.items()})
            out.loss.backward()
            torch.cuda.synchronize()
        finally:
            ops.TIMER = None
        res.append((out.loss.detach().clone(), acc.clone(), m.store.grad.clone()))
    for a, b_ in zip(*res):
        assert torch.equal(a, b_)


@pytest.mark.parametrize("case", ["padded", "left_pad_ragged", "varlen_packed", "chunked", "lora_all"])
def test_lm_head_over_label_rows_equals_every_row(dev, monkeypatch, case):
    """training steps run lm_head / cross entropy / lm_head dX over the labelled rows only (LM_HEAD_LABEL_ROWS): the loss equals the
    every-row form to fp32 summation order (1e-6), the accuracy exactly, every gradient to bf16 product tiling (the same rows go
    through the same kernels; cosine >= 0.99999, max |diff| <= 4.5e-3 max|g|), on padded, ragged left-padded, packed (varlen) batches
    and with the head chunked into several row blocks.  Three forms: every row, head over the labelled rows, head AND everything
    behind the last layer's attention over the labelled rows (LAST_LAYER_LABEL_ROWS; "lora_all": adapters on all seven projections,
    so o / gate / up / down of the last layer take their LoRA gradients from the selected rows).  The eval forward (labels + logits
    out) keeps every row."""
    from slam_llm_amd import model as model_mod
    cfg = dict(O.make_config(), lora_dropout=0.0, varlen=(case == "varlen_packed"))
    if case == "lora_all":
        cfg["lora_targets"] = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")
    W = O.init_weights(cfg, seed=42)
    audio = O.synth_audio(3, 1.0, seed=31)
    ob = O.synth_batch(cfg, audio, prompt_len=5, answer_lens=(4, 9, 6) if case != "padded" else (7,), seed=32,
                       left_pad=(case == "left_pad_ragged"), pad_to_30s=False)
    gb = {k: v.to(dev) for k, v in ob.items()}
    n_lab = int((ob["labels"][:, 1:] != -100).sum())
    assert 0 < n_lab < ob["labels"].numel() // 2
    res = []
    for on, last in ((False, False), (True, False), (True, True)):
        monkeypatch.setattr(model_mod, "LM_HEAD_LABEL_ROWS", on)
        monkeypatch.setattr(model_mod, "LAST_LAYER_LABEL_ROWS", last)
        m = model_mod.SlamHipModel(dict(cfg), dev).load_weights(W)
        m.train()
        if case == "chunked":
            m.llm.lm_head_chunk_rows = 8
        shapes = []
        orig = model_mod.ops.ce_fwd_bwd

        def spy(lg, *a, **k):
            shapes.append(lg.shape[0])
            return orig(lg, *a, **k)
        monkeypatch.setattr(model_mod.ops, "ce_fwd_bwd", spy)
        out, acc = m(**{k: v.clone() for k, v in gb.items()})
        out.loss.backward()
        monkeypatch.setattr(model_mod.ops, "ce_fwd_bwd", orig)
        rows_total = ob["labels"].numel() if case != "varlen_packed" else int(ob["attention_mask"].sum())
        assert sum(shapes) == (n_lab if on else rows_total), (shapes, n_lab)
        res.append((float(out.loss.detach()), float(acc), m.store.grad.clone()))
        if last:
            assert m.llm._last_pruned_rows == n_lab
        if on:      # eval forward with labels: full logits come back, every row is computed
            m.eval()
            with torch.no_grad():
                eo, _ = m(**{k: v.clone() for k, v in gb.items()})
            assert eo.logits.shape[:2] == ob["input_ids"].shape and abs(float(eo.loss) - res[0][0]) < 1e-4
    l0, a0, g0 = res[0]
    for l1, a1, g1 in res[1:]:
        assert abs(l0 - l1) <= 1e-6 * max(1.0, abs(l0)) and a0 == a1, (l0, l1, a0, a1)
        cs = float((g0 * g1).sum() / (g0.norm() * g1.norm()))
        rel = float((g0 - g1).abs().max()) / float(g0.abs().max())
        print(f"label-rows vs every-row [{case}]: 1 - cos {1 - cs:.2e}, max |diff| / max |g| {rel:.2e}")
        # measured on MI355X (round 5, `pytest -s`): rel 0 (packed, chunked head alone), 1.2e-3 .. 2.1e-3 where the last layer runs behind
        # its attention over the labelled rows only (other row counts -> other GEMM forms -> another fp32 summation order; ONE bf16 ulp of
        # the largest gradient element is 3.9e-3 of it).  Bound = 2 x the largest measured value and ~1 ulp; 1 - cos measured <= 2.9e-6.
        assert cs >= 0.99999 and rel <= 4.5e-3, (cs, rel)


@pytest.mark.parametrize("ragged", [False, True])
def test_unfrozen_hubert_encoder_matches_oracle(dev, ragged):
    _unfrozen_wave_encoder_case(dev, "hubert", ragged)


@pytest.mark.parametrize("which", ["hubert", "wavlm"])
def test_unfrozen_base_geometry_encoder_matches_oracle(dev, which):
    """row f4, base geometries (HuBERT Base, WavLM Base / Base+): conv -> GroupNorm over time -> GELU on layer 0 (adjoint:
    slam_groupnorm_time_gelu_bwd on the fp32 conv output), conv -> GELU after it, no conv bias; post-LN layers behind the encoder
    LayerNorm (x = LN1(x + attn(x)); x = LN2(x + ffn(x))), WavLM's gate reading the un-normalised attention input.  Ragged batch."""
    _unfrozen_wave_encoder_case(dev, which, True, base=True)


@pytest.mark.parametrize("ragged", [False, True])
def test_unfrozen_wavlm_encoder_matches_oracle(dev, ragged):
    """row f4, WavLM-Large's graph (models/slam_model.py:110-113 + :333-334): HuBERT-large's adjoint plus (1) the gated relative
    position bias inside the attention backward -- dL/dscore recomputed with the bias, d_gate[b,h,q] = sum_k dS * table[k - q],
    d_table[h][k - q] += sum_b,q gate * dS --, (2) the gate's own adjoint through the two sigmoids into grep_linear / grep_a and the
    attention input, (3) the bucket embedding's gradient gathered from d_table, summed over ALL layers (they share layer 0's table),
    (4) the chain rule of weight_norm on the positional conv (weight_g / weight_v stay the parameters).  mask_emb is a parameter that
    extract_features(mask=False) never reads: autograd leaves its grad None, the flat buffer keeps zeros."""
    _unfrozen_wave_encoder_case(dev, "wavlm", ragged)


@pytest.mark.parametrize("tag", ["A", "B", "C", "D", "E"])
def test_unfrozen_wavlm_train_mode_regularisers_match_oracle_with_the_same_masks(dev, tag):
    """VERDICT r3 missing #4: the un-frozen WavLM is left in TRAIN mode by the reference (slam_model.py:317-318): dropout_input, the
    dropout after the positional conv, attention_dropout inside the attention kernels (on top of the gated bias), dropout1 / 2 / 3 of
    every layer and encoder_layerdrop (WavLM.py:180-185, 353, 584, 596-597, 702-726).  The HIP encoder draws counter-based masks and
    numpy-stream layerdrop decisions; the masks it drew are rebuilt from their keys / seeds, the kept-layer pattern is read from its
    stash, both go to the fp32 oracle (itself pinned against the reference module in .train(), tests/golden/wavlm_train_tiny.npz), and
    output + every parameter gradient must agree.  A: layer 1 skipped, ragged batch; B: layer 0 skipped (no position bias anywhere,
    the gates' and the bucket table's gradients are exactly zero); C: nothing skipped; D / E: the Base structure (post-LN layers: the
    dropout after the positional conv sits behind the encoder LayerNorm, dropout1 / 3 in front of the LayerNorms), ragged batch all
    kept / layer 1 skipped.  Then eval mode: no key is drawn and the output equals the deterministic graph's; p = 0 in train mode:
    bit-identical to eval."""
    _wavlm_train_mode_case(dev, tag)


def _wavlm_train_mode_case(dev, tag, mask_seed=None, report=None):
    from oracle.make_golden_cases import WAVLM_BASE_TINY, WAVLM_TRAIN_TINY
    from slam_llm_amd.model import HipWavLMEncoder
    C, wseed = (WAVLM_TRAIN_TINY, 9) if tag in "ABC" else (WAVLM_BASE_TINY, 10)
    fx = G.load("wavlm_train_tiny")
    pattern = tuple(bool(k) for k in fx[tag + ".kept"])
    _wave_train_mode_case(dev, tag, HipWavLMEncoder, C, O.init_wavlm_weights(C, seed=wseed), O.wavlm_encoder, torch.from_numpy(fx[tag + ".wav"]),
                          [int(x) for x in fx[tag + ".n_valid"]], pattern, torch.from_numpy(fx[tag + ".cot"]), {}, True, 1 + 19 * pattern.count(False),
                          mask_seed=mask_seed, report=report)


@pytest.mark.parametrize("pattern,ragged", [((True, True), True), ((True, False), False)])
def test_unfrozen_hubert_train_mode_regularisers_match_oracle_with_the_same_masks(dev, pattern, ragged):
    """the same regularisers on the un-frozen HuBERT (fairseq's wav2vec2 TransformerEncoder: the module tree WavLM.py vendors, minus the
    position bias): fairseq parameter names and (g, v) of the positional conv in the store, the oracle's HF-named twin reached through the
    key map; masks rebuilt from the keys, kept-layer pattern from numpy's stream (the reference module itself is un-vendored fairseq:
    the sites are the ones pinned for WavLM)"""
    from oracle.make_golden_cases import HUBERT_TINY as C
    from slam_llm_amd.model import HipHubertEncoder
    W = O.init_hubert_weights(C, seed=7, weight_norm=True)
    N = 16000
    wav = torch.nn.functional.layer_norm(O.synth_audio(2, 1.0, seed=33), (N,))
    nv = [N, 11200] if ragged else [N, N]
    if ragged:
        wav[1, nv[1]:] = 0.0
    T = O.hubert_encoder(W, C, wav[:1, :N]).shape[1]
    cot = torch.randn((2, T, C["hub_dim"]), generator=torch.Generator().manual_seed(5)) * 0.1
    cot = cot.masked_fill(O.hubert_frame_padding_mask(N, T, torch.tensor(nv))[:, :, None], 0.0)
    probe = HipHubertEncoder(dict(C), dev, store=__import__("slam_llm_amd.model", fromlist=["TrainableStore"]).TrainableStore(dev))
    kmap = {hf: fs for fs, hf in probe.key_map("hf").items()}       # oracle (HF) name -> store (fairseq) name
    kmap["encoder.encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = "encoder.encoder.pos_conv.0.weight_g"
    kmap["encoder.encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = "encoder.encoder.pos_conv.0.weight_v"
    _wave_train_mode_case(dev, f"hubert {pattern}", HipHubertEncoder, C, W, O.hubert_encoder, wav, nv, pattern, cot, kmap, False,
                          16 * pattern.count(False))


# grep_linear.bias [8] and grep_a [H] of a WavLM layer are CANCELLING sums (sum_k dS = 0): their error is bounded against the norm of the
# layer's grep_linear.weight gradient.  Round 5: the RP dQ kernel derives Delta from the recomputed P (csrc/attention.hip, two passes), so
# sum_k dS is zero to fp32 rounding and what is left is the bf16 noise of the individual terms.  Measured over cases A..E x 6 mask draws
# on the GPU (tools/wavlm_trainmode_seeds.py -> profiles/r05_wavlm_trainmode_seeds.md): worst ratio 0.0262 (round 4, Delta from the
# bf16 O: 0.0505 at ONE draw against a bound of 0.05 -- the driver failure); worst cosine of a tensor-sized gate gradient 0.99587, worst
# norm deviation 0.0529.  Bounds = at least 2x the worst measured value: 0.06 / cosine floor 0.99 (1 - cos <= 2.4x) / norm 0.11.
SMALL_GATE_BOUND = 6.0e-2
TRAIN_MODE_NORM_TOL = 1.1e-1


def _wave_train_mode_case(dev, tag, enc_cls, C, W, oracle_fn, wav, nv, pattern, cot, kmap, has_relpos, min_zero, mask_seed=None, report=None):
    """mask_seed: torch seed the counter-based masks derive from (None: the suite-wide seed tests/conftest.py sets before every GPU
    test -- the draw no longer depends on which tests ran before).  report: a dict to fill with the measured errors instead of
    asserting the gradient bounds (tools/wavlm_trainmode_seeds.py: the bounds below are set from its table)."""
    from slam_llm_amd import ops
    from slam_llm_amd.model import TrainableStore
    if mask_seed is not None:
        torch.manual_seed(mask_seed)
    reg = dict(hub_dropout=0.1, hub_attention_dropout=0.1, hub_activation_dropout=0.1, hub_dropout_input=0.1, hub_layerdrop=0.4)
    store = TrainableStore(dev)
    cfg = dict(C, **reg)
    enc = enc_cls(cfg, dev, store=store)
    store.allocate()
    enc.bind()
    enc.load(W)
    store.refresh_bf16()
    enc.refresh()
    enc.train()
    ragged = any(n != wav.shape[1] for n in nv)
    np.random.seed(G.layerdrop_seed(pattern, 0.4))
    stash = {}
    out = enc.forward_train(wav.to(dev), stash, nv if ragged else None)
    S = stash["encoder"]
    B, T, d, H = out.shape[0], out.shape[1], C["hub_dim"], C["hub_heads"]
    kept = tuple(R is not None for R in S["blocks"])
    assert kept == pattern, (kept, pattern)
    ones = lambda n: torch.ones((B * T, n), dtype=torch.bfloat16, device=dev)      # noqa: E731
    hid = lambda key, n=d: (ops.dropout(ones(n), *key).float().cpu().view(B, T, n).ne(0).float() / 0.9)      # noqa: E731
    keys = [S["k_in"], S["k_x"]] + [R[k] for R in S["blocks"] if R is not None for k in ("k1", "k2", "k3")]
    assert all(k is not None for k in keys) and len({k[2] for k in keys}) == len(keys)
    tr = {"input": hid(S["k_in"]), "x": hid(S["k_x"]), "layers": []}
    Tp = (T + 63) // 64 * 64
    for R in S["blocks"]:
        if R is None:
            tr["layers"].append(None)
            continue
        am = torch.from_numpy(G.attn_keep_mask(R["ka"][1], 0.1, B, H, T, T, Tp, Tp)) / 0.9
        tr["layers"].append(dict(attn=am, d1=hid(R["k1"]), d2=hid(R["k2"], C["hub_ffn"]), d3=hid(R["k3"])))
        assert (R["rp"] is None) == (not pattern[0] or not has_relpos)
    frac = torch.cat([m.reshape(-1) for m in [tr["input"], tr["x"]] + [v for lm in tr["layers"] if lm for v in lm.values()]]).ne(0).float().mean()
    assert abs(float(frac) - 0.9) < 0.01, float(frac)
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    ref = oracle_fn(Wg, C, wav, n_valid=torch.tensor(nv) if ragged else None, train=tr)
    pad = O.hubert_frame_padding_mask(wav.shape[1], T, torch.tensor(nv))
    a = out.float().cpu().masked_fill(pad[:, :, None], 0.0).numpy()
    g = ref.detach().masked_fill(pad[:, :, None], 0.0).numpy()
    assert rel_err(a, g) < 3e-2 and G.cosine(g, a) > 0.9995, (rel_err(a, g), G.cosine(g, a))
    (ref * cot).sum().backward()
    enc.backward_hip(cot.to(dev).to(torch.bfloat16).reshape(B * T, d).contiguous(), stash, acc=False)
    gmax = max(float(v.grad.norm()) for v in Wg.values() if v.grad is not None)
    worst, n_zero, bad = (1.0, ""), 0, []
    for n in W:
        mine = store.grad_view(kmap.get(n, n)).float().cpu().reshape(W[n].shape)
        if Wg[n].grad is None or float(Wg[n].grad.abs().max()) == 0.0:     # the skipped layer, mask_emb, in B every gate + the bucket table
            assert float(mine.abs().max()) == 0.0, n
            n_zero += 1
            continue
        gold = Wg[n].grad
        if n.endswith("k_proj.bias") and float(gold.norm()) < 1e-4 * gmax:
            assert float(mine.abs().max()) < 3e-2, n
            continue
        cs = G.cosine(gold.numpy(), mine.numpy())
        worst = min(worst, (cs, n))
        # the conv stack sits under 3 layers of masked bf16 adjoints at 64-channel widths; the gate parameters are cancelling sums
        # (sum_k dS = 0), treated as in _unfrozen_wave_encoder_case below: tensor-sized ones by cosine 0.99, grep_linear.bias [8] and
        # grep_a [H = 2] by their error against the layer's grep_linear.weight gradient norm (measured: cosine 0.67 .. 0.9999 on the
        # 8-element bias from one mask draw to the next -- not a statistic)
        if ".grep_" in n and mine.numel() <= 16:
            wn = float(Wg[n.rsplit(".grep_", 1)[0] + ".grep_linear.weight"].grad.norm())
            ratio = float((mine - gold).norm()) / wn
            if report is not None:
                report.setdefault("small_gate_err_over_weight_grad_norm", {})[n] = ratio
            if ratio > SMALL_GATE_BOUND:
                bad.append((n, "err", float((mine - gold).norm()), "weight-grad norm", wn))
            continue
        floor = 0.99 if ("feature_extractor" in n or "grep_" in n) else 0.995
        nr = abs(float(mine.norm()) - float(gold.norm())) / float(gold.norm())
        if report is not None:
            report.setdefault("cos_norm", {})[n] = (cs, nr)
        if cs <= floor or nr >= TRAIN_MODE_NORM_TOL:
            bad.append((n, round(cs, 5), round(nr, 4)))
    if report is not None:
        report["bad"] = bad
        report["worst"] = worst
        return
    assert not bad, bad
    assert n_zero >= min_zero
    if os.environ.get("SLAM_TEST_VERBOSE"):
        print(f"train-mode {tag}: worst gradient cosine {worst}")
    # eval mode: nothing drawn; train mode at p = 0: the deterministic graph, bit for bit
    calls = enc._drop_calls
    enc.eval()
    st2 = {}
    out_eval = enc.forward_train(wav.to(dev), st2, nv if ragged else None)
    assert enc._drop_calls == calls and all(R is not None and R["ka"] is None and R["k1"] is None for R in st2["encoder"]["blocks"])
    enc.train()
    for k in reg:
        cfg[k] = 0.0
    out_p0 = enc.forward_train(wav.to(dev), {}, nv if ragged else None)
    assert enc._drop_calls == calls and torch.equal(out_p0, out_eval)


def _unfrozen_wave_encoder_case(dev, which, ragged, base=False):
    """row f4: train_config.freeze_encoder=false with the HuBERT encoder (models/slam_model.py:110-113 + :335-341) -- the hand-written
    adjoint of the whole graph: 7 conv layers (LayerNorm over channels + GELU; general col2im for k 10 / 3 / 2, strides 5 / 2), feature
    LayerNorm + projection, grouped positional conv (dX = the implicit-GEMM kernel on tap-reversed, channel-transposed weights; dW per
    group), pre-LN transformer layers with key biases.  Tiny widths (conv 64, d 128, 2 layers, pos conv k 16 in 4 groups), linear
    projector, ragged variant: second clip 70 % long (fairseq frame mask, padded frames zeroed before the positional conv and masked
    as keys).  Loss and EVERY gradient against the oracle's autograd: cosine >= 0.998, norm within 4 %."""
    from oracle.make_golden_cases import HUBERT_BASE_TINY, HUBERT_TINY, WAVLM_BASE_TINY, WAVLM_TINY
    from slam_llm_amd.model import SlamHipModel
    HUBERT_TINY = {("hubert", False): HUBERT_TINY, ("wavlm", False): WAVLM_TINY, ("hubert", True): HUBERT_BASE_TINY,
                   ("wavlm", True): WAVLM_BASE_TINY}[(which, base)]
    cfg = dict(O.make_config(), **HUBERT_TINY, lora_dropout=0.0)
    cfg.update(encoder_name=which, enc_dim=HUBERT_TINY["hub_dim"])
    W = {k: v for k, v in O.init_weights(cfg, seed=42).items() if not k.startswith("encoder.")}
    W.update(O.init_hubert_weights(HUBERT_TINY, seed=7, weight_norm=True) if which == "hubert" else O.init_wavlm_weights(HUBERT_TINY, seed=9))
    N = 16000
    wav = O.synth_audio(2, 1.0, seed=9)
    if not base:      # the large checkpoints' cfg has normalize=True (dataset-side layer norm of the waveform), the base ones have not
        wav = torch.nn.functional.layer_norm(wav, (N,))
    n_valid = [N, 11200] if ragged else [N, N]
    if ragged:
        wav[1, n_valid[1]:] = 0.0
    alen = [n // 320 // 5 for n in n_valid]                       # speech_dataset.py:98-99
    samples = [O.make_sample(alen[0], [5, 6, 7], [9, 10, 11, 12], 2), O.make_sample(alen[1], [5, 6], [9, 10], 2)]
    ob = O.collate_left_pad(samples, pad_id=2)
    names = O.trainable_names(W) + [n for n in W if n.startswith("encoder.")]
    for n in names:
        W[n].requires_grad_(True)
    enc = (O.hubert_encoder if which == "hubert" else O.wavlm_encoder)(W, cfg, wav, n_valid=torch.tensor(n_valid) if ragged else None)
    proj = O.projector_concat(W, enc, cfg["ds_rate"])
    emb = O.embed_splice(W["llm.base_model.model.model.embed_tokens.weight"], ob["input_ids"].clone(), ob["modality_mask"].bool(), proj)
    loss_ref, _ = O.llama_forward(W, cfg, emb, ob["attention_mask"], ob["labels"])
    loss_ref.backward()
    unused = [n for n in names if W[n].grad is None]
    assert unused == ([] if which == "hubert" else ["encoder.model.mask_emb"]), unused
    grads = {n: (W[n].grad.detach().clone() if W[n].grad is not None else torch.zeros_like(W[n])) for n in names}
    for n in names:
        W[n].requires_grad_(False)
        W[n].grad = None
    model = SlamHipModel(dict(cfg, freeze_encoder=False), dev).load_weights(W)
    model.train()
    # HuBERT trains under the names of the module the reference un-freezes (fairseq's HubertModel: ...conv_layers.N.0.weight, post_extract_proj,
    # encoder.pos_conv.0.weight_g / weight_v, self_attn.q_proj, ...); the oracle restates the HF twin: compare through the key map
    kmap = dict(model.encoder.key_map("hf")) if which == "hubert" else {}
    if which == "hubert":
        kmap["encoder.encoder.pos_conv.0.weight_g"] = "encoder.encoder.pos_conv_embed.conv.parametrizations.weight.original0"
        kmap["encoder.encoder.pos_conv.0.weight_v"] = "encoder.encoder.pos_conv_embed.conv.parametrizations.weight.original1"
        assert "encoder.post_extract_proj.weight" in model.store.params and "encoder.encoder.layers.0.self_attn.q_proj.bias" in model.store.params
    grads = {n: grads[kmap.get(n, n)] for n in model.store.params if kmap.get(n, n) in grads}
    assert len(grads) == len(names) and set(model.store.params) == set(grads), set(model.store.params) ^ set(grads)
    unused = [n for n in model.store.params if kmap.get(n, n) in unused]
    gb = {k: v.to(dev) for k, v in ob.items()}
    gb["audio"] = wav.to(dev)
    gb["audio_len"] = torch.tensor(n_valid, dtype=torch.int32, device=dev)

    class PrefixRecorder:      # a GradSync-shaped hook: every prefix it is handed must already hold its final values
        def __init__(self):
            self.seen = []

        def on_prefix(self, end):
            self.seen.append((end, model.store.grad[:end].clone()))
    rec = PrefixRecorder()
    model.grad_hooks.append(rec)
    outputs, _ = model(**gb)
    outputs.loss.backward()
    model.grad_hooks.remove(rec)
    assert len(rec.seen) >= 2 and rec.seen[-1][0] == model.store.size
    for end, snap in rec.seen:
        assert torch.equal(snap, model.store.grad[:end]), f"gradient prefix [0, {end}) was announced before it was final"
    assert abs(float(outputs.loss.detach()) - float(loss_ref.detach())) < 1.5e-2, (float(outputs.loss.detach()), float(loss_ref.detach()))
    gmax = max(float(v.norm()) for v in grads.values())
    worst, worst_name = 1.0, ""
    for n, p in model.store.params.items():
        gn, mine = float(grads[n].norm()), p.grad.float().cpu()
        assert mine.shape == grads[n].shape, n
        if n in unused:
            assert float(mine.abs().max()) == 0.0, n
            continue
        if n.endswith("k_proj.bias") and gn < 1e-4 * gmax:          # key biases: mathematically zero gradient
            assert float(mine.abs().max()) < 3e-2, n
            continue
        cs = G.cosine(grads[n].numpy(), mine.numpy())
        if cs < worst:
            worst, worst_name = cs, n
        # the conv feature extractor sits under 2 transformer layers, the positional conv and up to 7 bf16 conv / LayerNorm adjoints
        # (64-channel rows at these widths): 0.995 there (measured 0.9971 on conv_layers.0), 0.998 everywhere else
        floor = G.FLOORS["unfrozen_fe"] if "feature_extractor" in n else G.FLOORS["unfrozen"]
        if base and which == "wavlm":
            # WavLM Base at these widths is the noisiest case of the family: the projector's linear1 gradient -- which does not pass through
            # the encoder backward at all, only through the bf16 forward and the LLM backward -- already sits at 0.9978 here, and the
            # encoder's own gradients follow it uniformly (measured 0.9948 worst, layer 1's key projection; norms within 4 %)
            floor = G.FLOORS["unfrozen_wavlm_base"]
        if ".grep_" in n or "relative_attention_bias" in n:
            # WavLM's gate / bias-table parameters: d(gate)[q] = sum_k dS[q,k] table[k - q] is a cancelling sum (sum_k dS = 0), so the ~5 %
            # bf16 noise dS carries in this tiny end-to-end case (the same noise that puts q / k weights at 0.9985) is amplified; the
            # kernels themselves are pinned on clean inputs at 0.999 / 0.9999 (test_attn_bwd_with_gated_relative_position_bias,
            # test_wavlm_gate_backward_matches_autograd).  Tensor-sized ones: 0.99 (measured 0.9912 worst).  grep_linear.bias [8] and
            # grep_a [H = 2] are column sums of the per-frame terms whose un-summed form is grep_linear.weight's gradient: their error
            # is bounded against THAT scale (5 %), a cosine over 2 to 8 elements of a near-cancelling sum is not a statistic.
            if mine.numel() <= 16:
                wn = float(grads[n.rsplit(".grep_", 1)[0] + ".grep_linear.weight"].norm())
                err = float((mine - grads[n]).norm())
                if os.environ.get("SLAM_TEST_VERBOSE"):
                    print(f"  {n:80s} err {err:.3e} vs weight-grad norm {wn:.3e}")
                assert err <= SMALL_GATE_BOUND * wn, f"grad {n}: error {err} vs {SMALL_GATE_BOUND} of the layer's grep_linear.weight gradient norm {wn}"
                continue
            floor = G.FLOORS["unfrozen_gate"]
        if os.environ.get("SLAM_TEST_VERBOSE"):     # prints only: the asserts below run either way
            print(f"  {n:80s} cos {cs:.5f}  norm {float(mine.norm()):.4e} vs {gn:.4e}")
        G.floor_check(cs, floor, f"grad {n}: cosine {cs}")
        assert abs(float(mine.norm()) - gn) <= 5e-2 * gn + 1e-7, f"grad {n}: norm {float(mine.norm())} vs {gn}"
    print(f"unfrozen {which} (ragged={ragged}): worst gradient cosine {worst:.6f} ({worst_name})")
    # one optimizer step moves the encoder, and the state_dict carries it under the reference's names
    from slam_llm_amd.model import SlamAdamW
    before = model.store.flat.clone()
    opt = SlamAdamW(model, lr=1e-3)
    opt.step()
    opt.zero_grad()
    assert float((model.store.flat - before).abs().max()) > 0
    sd = model.state_dict()
    assert ("encoder.feature_extractor.conv_layers.0.0.weight" if which == "hubert" else "encoder.model.encoder.pos_conv.0.weight_g") in sd
    if which == "hubert":      # a checkpoint written from named_parameters (save_model_checkpoint_peft) loads back, and so does the HF-named W
        assert "encoder.encoder.pos_conv.0.weight_v" in sd and not any("pos_conv_embed" in k for k in sd)
        again = SlamHipModel(dict(cfg, freeze_encoder=False), dev).load_weights({**{k: v for k, v in W.items() if not k.startswith("encoder.")},
                                                                                  **{k: v.detach().cpu() for k, v in sd.items() if k.startswith("encoder.")}})
        for k_, v_ in again.store.params.items():      # (the optimizer step above moved the adapters / projector too: W still holds their old values)
            if k_.startswith("encoder."):
                assert torch.equal(v_, model.store.params[k_]), k_
    out2, _ = model(**gb)
    assert bool(torch.isfinite(out2.loss))


@pytest.mark.parametrize("projector", ["cov1d-linear", "q-former"])
def test_unfrozen_whisper_with_cov1d_and_qformer_projectors(dev, projector):
    """row f4: train_config.freeze_encoder=false (models/slam_model.py:110-113 is projector-agnostic) with EncoderProjectorCov1d /
    EncoderProjectorQFormer in front of the trainable Whisper encoder -- the projector backward now hands dL/d(encoder output) on
    (cov1d: dc . Wc un-stacked; Q-Former: sum over its cross-attention layers of d[k | v] . [Wk ; Wv]).  Loss and EVERY gradient
    (encoder, projector, LoRA) against the oracle's autograd on the same batch: cosine >= 0.998, norm within 4 % (the Q-Former
    fixtures' bound), zero-gradient key biases exempt as in test_qformer_*."""
    from slam_llm_amd.model import SlamHipModel
    cfg = dict(O.make_config(), lora_dropout=0.0)
    W = {k: v for k, v in O.init_weights(cfg, seed=42).items() if not k.startswith("encoder_projector.")}
    extra = {}
    if projector == "q-former":
        extra = O.qformer_config(qf_layers=2, qf_queries=8)
        W.update(O.init_qformer_weights(extra, cfg["enc_dim"], cfg["llm_dim"], seed=11))
    else:
        W.update(O.init_cov1d_weights(cfg["enc_dim"], cfg["llm_dim"], cfg["ds_rate"], hidden=cfg["proj_hidden"], seed=13))
    audio = O.synth_audio(2, 2.0, seed=1234)
    mels = [O.log_mel_spectrogram(a[: a.shape[0] // 160 * 160], cfg["n_mels"]).permute(1, 0) for a in audio]
    T2 = (mels[0].shape[0] + 1) // 2
    alen = extra["qf_queries"] if projector == "q-former" else T2 // cfg["ds_rate"]
    g = torch.Generator().manual_seed(1236)
    samples = [O.make_sample(alen, torch.randint(3, cfg["vocab"], (6,), generator=g).tolist(),
                             torch.randint(3, cfg["vocab"], (al - 1,), generator=g).tolist(), 2) for al in (5, 9)]
    ob = O.collate_left_pad(samples, pad_id=2, mels=mels)
    names = O.trainable_names(W) + [n for n in W if n.startswith("encoder.") and not n.endswith("positional_embedding")]
    for n in names:
        W[n].requires_grad_(True)
    enc = O.whisper_encoder(W, cfg, ob["audio_mel"].permute(0, 2, 1))
    proj = (O.projector_qformer(W, extra, enc, ob["audio_mel_post_mask"]) if projector == "q-former"
            else O.projector_cov1d(W, enc, cfg["ds_rate"]))
    emb = O.embed_splice(W["llm.base_model.model.model.embed_tokens.weight"], ob["input_ids"].clone(), ob["modality_mask"].bool(), proj)
    loss_ref, _ = O.llama_forward(W, cfg, emb, ob["attention_mask"], ob["labels"])
    loss_ref.backward()
    grads = {n: W[n].grad.detach().clone() for n in names}
    for n in names:
        W[n].requires_grad_(False)
        W[n].grad = None
    model = SlamHipModel(dict(cfg, **extra, projector=projector, freeze_encoder=False, qf_dropout=0.0), dev).load_weights(W)
    model.train()
    assert set(model.store.params) == set(names)
    outputs, _ = model(**{k: v.to(dev) for k, v in ob.items()})
    outputs.loss.backward()
    assert abs(float(outputs.loss.detach()) - float(loss_ref.detach())) < 1.5e-2, (float(outputs.loss.detach()), float(loss_ref.detach()))
    gmax = max(float(v.norm()) for v in grads.values())
    worst = 1.0
    for n, p in model.store.params.items():
        gn, mine = float(grads[n].norm()), p.grad.float().cpu()
        if n.endswith("key.bias") and gn < 1e-4 * gmax:
            assert float(mine.abs().max()) < 3e-2, n
            continue
        cs = G.cosine(grads[n].numpy(), mine.numpy())
        worst = min(worst, cs)
        G.floor_check(cs, G.FLOORS["unfrozen"], f"grad {n}: cosine {cs}")
        assert abs(float(mine.norm()) - gn) <= 4e-2 * gn + 1e-7, f"grad {n}: norm {float(mine.norm())} vs {gn}"
    assert sum(1 for n in model.store.params if n.startswith("encoder.")) == 4 + 15 * cfg["enc_layers"] + 2
    print(f"unfrozen whisper + {projector}: worst gradient cosine {worst:.6f}")


def test_matches_oracle_with_gpu_logmel(dev):
    """raw audio in -> GPU log-mel inside the step; oracle computes the same on the CPU."""
    cfg = CASES["step_tiny"]["cfg"]
    model, W = build(cfg, dev)
    model.train()
    audio = O.synth_audio(2, 1.3, seed=7)
    ob = O.synth_batch(cfg, audio, prompt_len=5, answer_lens=(4, 6), seed=3, left_pad=True, pad_to_30s=True)
    with torch.no_grad():
        loss_ref, _, acc_ref, aux = O.slam_forward(W, cfg, ob)
    gb = {k: v.to(dev) for k, v in ob.items() if k != "audio_mel"}
    gb["audio"] = audio.to(dev)
    outputs, acc = model(**gb)
    assert abs(float(outputs.loss) - float(loss_ref)) < 1e-2


def test_grad_accumulation_and_foreign_optimizer(dev):
    """two backward passes accumulate; torch.optim.AdamW over model.parameters() also drives the HIP path."""
    cfg = CASES["step_tiny"]["cfg"]
    fx = G.load("step_tiny")
    model, W = build(cfg, dev)
    model.train()
    b = batch_from_fixture(fx, dev)
    out1, _ = model(**{k: v.clone() for k, v in b.items()})
    out1.loss.backward()
    g1 = model.store.grad.clone()
    out2, _ = model(**{k: v.clone() for k, v in b.items()})
    (out2.loss * 0.5).backward()
    assert torch.allclose(model.store.grad, 1.5 * g1, rtol=2e-2, atol=1e-6)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    opt.step()
    opt.zero_grad()
    out3, _ = model(**{k: v.clone() for k, v in b.items()})
    assert float(out3.loss) < float(out1.loss)


def test_hubert_encoder_matches_reference_twin_fixture(dev):
    """a11: HuBERT encoder forward vs the HF-HubertModel fixture and the oracle (bf16 path: <= 3e-2 of tensor scale)"""
    from oracle.make_golden_cases import HUBERT_TINY
    from slam_llm_amd.model import HipHubertEncoder
    fx = G.load("hubert_tiny")
    W = O.init_hubert_weights(HUBERT_TINY, seed=7)
    enc = HipHubertEncoder(dict(HUBERT_TINY), dev).load(W)
    wav = torch.from_numpy(fx["wav"]).to(dev)
    out = enc.forward_wav(wav)
    assert list(out.shape) == [int(x) for x in fx["out_shape"]]
    assert enc.out_frames(wav.shape[1]) == out.shape[1]
    g, a = G.sub(fx, "out", out.float().cpu().numpy())
    assert rel_err(a, g) < 3e-2, f"hubert out rel err {rel_err(a, g)}"
    assert G.cosine(g, a) > 0.9995
    assert HUBERT_TINY["hub_conv_kernel"] == (10, 3, 3, 3, 3, 2, 2)
    # 30 s of 16 kHz audio -> 1499 frames (one fewer than the 300*5 slots the dataset reserves)
    assert enc.out_frames(480000) == 1499


def test_hubert_linear_llm_step_matches_oracle(dev):
    """alt-encoder path end to end (HuBERT -> linear projector -> LLM+LoRA): loss vs the CPU oracle"""
    from oracle.make_golden_cases import HUBERT_TINY
    from slam_llm_amd.model import SlamHipModel
    cfg = dict(O.make_config(), **HUBERT_TINY)
    cfg.update(encoder_name="hubert", enc_dim=HUBERT_TINY["hub_dim"])
    W = O.init_weights(cfg, seed=42)
    W = {k: v for k, v in W.items() if not k.startswith("encoder.")}
    W.update(O.init_hubert_weights(HUBERT_TINY, seed=7))
    model = SlamHipModel(dict(cfg, lora_dropout=0.0, qf_dropout=0.0), dev).load_weights(W)
    model.train()
    wav = torch.nn.functional.layer_norm(O.synth_audio(2, 1.0, seed=9), (16000,))
    with torch.no_grad():
        enc = O.hubert_encoder(W, cfg, wav)
        proj = O.projector_concat(W, enc, cfg["ds_rate"])
    Ta = proj.shape[1]
    samples = [O.make_sample(Ta + 1, [5, 6, 7], [9, 10, 11, 12], 2), O.make_sample(Ta + 1, [5, 6], [9, 10], 2)]
    ob = O.collate_left_pad(samples, pad_id=2)
    with torch.no_grad():
        emb = O.embed_splice(W["llm.base_model.model.model.embed_tokens.weight"], ob["input_ids"].clone(), ob["modality_mask"].bool(), proj)
        loss_ref, _ = O.llama_forward(W, cfg, emb, ob["attention_mask"], ob["labels"])
    gb = {k: v.to(dev) for k, v in ob.items()}
    gb["audio"] = wav.to(dev)
    gb["audio_mask"] = torch.ones_like(gb["audio"])
    outputs, acc = model(**gb)
    assert abs(float(outputs.loss.detach()) - float(loss_ref)) < 1e-2
    outputs.loss.backward()
    assert torch.isfinite(model.store.grad).all() and float(model.store.grad.abs().sum()) > 0


def test_wavlm_encoder_matches_reference_fixture(dev):
    """f4: HipWavLMEncoder (HuBERT-large graph + gated relative position bias inside the attention kernel) vs the fixture written
    by the reference's own WavLM module: equal-length batch and ragged zero-padded batch (valid frames)"""
    from oracle.make_golden_cases import WAVLM_TINY as C
    from slam_llm_amd.model import HipWavLMEncoder
    fx = G.load("wavlm_tiny")
    W = O.init_wavlm_weights(C, seed=9)
    enc = HipWavLMEncoder(dict(C), dev).load(W)
    out = enc.forward_wav(torch.from_numpy(fx["wav"]).to(dev)).float().cpu().numpy()
    assert list(out.shape) == [int(x) for x in fx["out_shape"]]
    g, a = G.sub(fx, "out", out)
    assert rel_err(a, g) < 3e-2 and G.cosine(g, a) > 0.9995, (rel_err(a, g), G.cosine(g, a))
    nv = [int(x) for x in fx["ragged.n_valid"]]
    out_r = enc.forward_wav(torch.from_numpy(fx["ragged.wav"]).to(dev), nv).float().cpu()
    pad = torch.from_numpy(fx["ragged.frame_padding_mask"])
    g, a = G.sub(fx, "ragged.out", out_r.masked_fill(pad[:, :, None], 0.0).numpy())
    assert rel_err(a, g) < 3e-2 and G.cosine(g, a) > 0.9995, (rel_err(a, g), G.cosine(g, a))
    # the bias matters: without it the output is far from the fixture (guards against a silently ignored table)
    enc.w["rel_bias"].zero_()
    enc._tables = {}
    off = enc.forward_wav(torch.from_numpy(fx["wav"]).to(dev)).float().cpu().numpy()
    g, a = G.sub(fx, "out", off)
    assert rel_err(a, g) > 6e-2


def test_hubert_base_encoder_matches_hf_fixture(dev):
    """HuBERT-base structure through the HF state-dict names (no conv bias, GroupNorm on conv layer 0 only, post-LN layers) vs the
    fixture written by HF HubertModel: equal-length batch and ragged zero-padded batch (valid frames)"""
    from oracle.make_golden_cases import HUBERT_BASE_TINY as C
    from slam_llm_amd.model import HipHubertEncoder
    fx = G.load("hubert_base_tiny")
    W = O.init_hubert_weights(C, seed=8)
    enc = HipHubertEncoder(dict(C), dev).load(W)
    out = enc.forward_wav(torch.from_numpy(fx["wav"]).to(dev)).float().cpu().numpy()
    assert list(out.shape) == [int(x) for x in fx["out_shape"]]
    g, a = G.sub(fx, "out", out)
    assert rel_err(a, g) < 3e-2 and G.cosine(g, a) > 0.9995, (rel_err(a, g), G.cosine(g, a))
    nv = [int(x) for x in fx["ragged.n_valid"]]
    out_r = enc.forward_wav(torch.from_numpy(fx["ragged.wav"]).to(dev), nv).float().cpu()
    pad = torch.from_numpy(fx["ragged.frame_padding_mask"])
    g, a = G.sub(fx, "ragged.out", out_r.masked_fill(pad[:, :, None], 0.0).numpy())
    assert rel_err(a, g) < 3e-2 and G.cosine(g, a) > 0.9995, (rel_err(a, g), G.cosine(g, a))


def test_wavlm_base_encoder_matches_reference_fixture(dev):
    """f4: the Base / Base+ structure (GroupNorm over time after the first conv via slam_groupnorm_time_gelu, conv -> GELU for the
    other layers, post-LN layers behind the encoder-level LayerNorm) vs the fixture written by the reference's own WavLM module
    configured that way: equal-length batch and ragged zero-padded batch (valid frames)"""
    from oracle.make_golden_cases import WAVLM_BASE_TINY as C
    from slam_llm_amd.model import HipWavLMEncoder
    fx = G.load("wavlm_base_tiny")
    W = O.init_wavlm_weights(C, seed=10)
    enc = HipWavLMEncoder(dict(C), dev).load(W)
    out = enc.forward_wav(torch.from_numpy(fx["wav"]).to(dev)).float().cpu().numpy()
    assert list(out.shape) == [int(x) for x in fx["out_shape"]]
    g, a = G.sub(fx, "out", out)
    assert rel_err(a, g) < 3e-2 and G.cosine(g, a) > 0.9995, (rel_err(a, g), G.cosine(g, a))
    nv = [int(x) for x in fx["ragged.n_valid"]]
    out_r = enc.forward_wav(torch.from_numpy(fx["ragged.wav"]).to(dev), nv).float().cpu()
    pad = torch.from_numpy(fx["ragged.frame_padding_mask"])
    g, a = G.sub(fx, "ragged.out", out_r.masked_fill(pad[:, :, None], 0.0).numpy())
    assert rel_err(a, g) < 3e-2 and G.cosine(g, a) > 0.9995, (rel_err(a, g), G.cosine(g, a))


def test_groupnorm_over_time_with_gelu(dev):
    """slam_groupnorm_time_gelu == F.gelu(F.group_norm(x, C groups)) on [B, C, T] (one group per channel, statistics over time),
    T not a multiple of the kernel's row chunk, a large offset to exercise the variance (fp64 combination of the partial sums)"""
    from slam_llm_amd import ops
    B, T, C = 3, 1000, 192
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(B * T, C, generator=g, device=dev) * 2.0 + 30.0
    wgt = 1 + 0.1 * torch.randn(C, generator=g, device=dev)
    bias = 0.1 * torch.randn(C, generator=g, device=dev)
    y = ops.groupnorm_time_gelu(x, B, T, wgt, bias, 1e-5).float()
    ref = torch.nn.functional.gelu(torch.nn.functional.group_norm(x.view(B, T, C).transpose(1, 2), C, wgt, bias, 1e-5)).transpose(1, 2)
    err = (y.view(B, T, C) - ref).abs().max()
    assert float(err) < 2e-2, float(err)      # bf16 output rounding of O(1) values


@pytest.mark.parametrize("case", ["large", "base"])
def test_wavlm_llm_step_matches_oracle(dev, case):
    """WavLM (Large structure / Base structure) -> linear projector -> LLM + LoRA training step at tiny widths: loss vs the oracle,
    and it trains"""
    from oracle.make_golden_cases import WAVLM_BASE_TINY, WAVLM_TINY
    from slam_llm_amd.model import SlamAdamW, SlamHipModel
    WAVLM_TINY = WAVLM_TINY if case == "large" else WAVLM_BASE_TINY
    cfg = dict(O.make_config(), **WAVLM_TINY)
    cfg.update(encoder_name="wavlm", enc_dim=WAVLM_TINY["hub_dim"])
    W = {k: v for k, v in O.init_weights(cfg, seed=42).items() if not k.startswith("encoder.")}
    W.update(O.init_wavlm_weights(WAVLM_TINY, seed=9))
    model = SlamHipModel(dict(cfg, lora_dropout=0.0), dev).load_weights(W)
    model.train()
    wav = O.synth_audio(2, 1.0, seed=9)
    if case == "large":     # the Large checkpoint's cfg has normalize=True (dataset-side layer norm of the waveform), Base has not
        wav = torch.nn.functional.layer_norm(wav, (16000,))
    Ta = 49 // cfg["ds_rate"]
    samples = [O.make_sample(Ta, [5, 6, 7], [9, 10, 11, 12], 2), O.make_sample(Ta, [5, 6], [9, 10], 2)]
    ob = O.collate_left_pad(samples, pad_id=2)
    with torch.no_grad():
        enc = O.wavlm_encoder(W, cfg, wav)
        proj = O.projector_concat(W, enc, cfg["ds_rate"])
        emb = O.embed_splice(W["llm.base_model.model.model.embed_tokens.weight"], ob["input_ids"].clone(), ob["modality_mask"].bool(), proj)
        loss_ref, _ = O.llama_forward(W, cfg, emb, ob["attention_mask"], ob["labels"])
    gb = {k: v.to(dev) for k, v in ob.items()}
    gb["audio"] = wav.to(dev)
    opt = SlamAdamW(model, lr=1e-3)
    losses = []
    for _ in range(3):
        outputs, acc = model(**{k: v.clone() for k, v in gb.items()})
        outputs.loss.backward()
        opt.step(); opt.zero_grad()
        losses.append(float(outputs.loss.detach()))
    assert abs(losses[0] - float(loss_ref)) < 1.5e-2, (losses[0], float(loss_ref))
    assert losses[2] < losses[0]


def test_qformer_projector_matches_reference_fixture(dev):
    """a3': Q-Former projector forward + every parameter gradient vs the reference module's fixture"""
    from oracle.make_golden_cases import QFORMER_CASE as C
    from slam_llm_amd.model import TrainableStore
    from slam_llm_amd.qformer import HipProjectorQFormer
    fx = G.load("qformer")
    cfg = dict(C["cfg"], enc_dim=C["enc_dim"], llm_dim=C["llm_dim"], qf_dropout=0.0)   # the fixture is the eval-mode module
    W = O.init_qformer_weights(C["cfg"], C["enc_dim"], C["llm_dim"], seed=11)
    store = TrainableStore(dev)
    qf = HipProjectorQFormer(cfg, store)
    store.allocate()
    qf.bind()
    with torch.no_grad():
        for n, p in store.params.items():
            p.copy_(W[n].to(dev))
    assert set(store.params) == set(W)
    store.refresh_bf16()
    qf.refresh()
    x = torch.from_numpy(fx["x"]).to(dev).to(torch.bfloat16)
    atts = torch.from_numpy(fx["atts"]).to(dev)
    stash = {}
    out = qf.forward_hip(x, atts, stash)
    g, a = G.sub(fx, "out", out.float().cpu().numpy())
    assert rel_err(a, g) < 3e-2 and G.cosine(g, a) > 0.9995, (rel_err(a, g), G.cosine(g, a))
    cot = torch.from_numpy(fx["cot"]).to(dev).to(torch.bfloat16).reshape(-1, C["llm_dim"]).contiguous()
    qf.backward_hip(cot, stash, acc=False)
    worst = 1.0
    for n in W:
        gold, mine = G.sub(fx, "grad." + n, store.grad_view(n).float().cpu().numpy())
        gn = float(fx["grad." + n + ".__norm"])
        if gn < 1e-4:  # key biases: mathematically zero gradient
            assert np.abs(mine).max() < 3e-2, n  # bf16 rounding noise of the dK column sum
            continue
        cs = G.cosine(gold, mine)
        worst = min(worst, cs)
        G.floor_check(cs, 0.998, f"grad {n}: cosine {cs}")
        mn = float(np.sqrt((store.grad_view(n).float().cpu().numpy().astype(np.float64) ** 2).sum()))
        assert abs(mn - gn) < 4e-2 * gn, f"grad {n}: norm {mn} vs {gn}"
    # state-dict keys equal the reference module's
    assert {k for k in qf.state_dict()} == {k[len("encoder_projector."):] for k in W}


def test_qformer_train_mode_dropout_matches_oracle_with_the_same_masks(dev):
    """train mode: the four kinds of hidden dropout (query LayerNorm output; self / cross / feed-forward output projections) and
    the dropout on the attention probabilities of every self- / cross-attention -- the masks the kernels drew are rebuilt from
    their keys / seeds and handed to the fp32 oracle; output and every parameter gradient must agree as in the eval-mode test"""
    from oracle.make_golden_cases import QFORMER_CASE as C
    from slam_llm_amd import ops
    from slam_llm_amd.model import TrainableStore
    from slam_llm_amd.qformer import HipProjectorQFormer
    fx = G.load("qformer")
    cfg = dict(C["cfg"], enc_dim=C["enc_dim"], llm_dim=C["llm_dim"], qf_dropout=0.1)
    W = O.init_qformer_weights(C["cfg"], C["enc_dim"], C["llm_dim"], seed=11)
    store = TrainableStore(dev)
    qf = HipProjectorQFormer(cfg, store)
    store.allocate()
    qf.bind()
    with torch.no_grad():
        for n, p in store.params.items():
            p.copy_(W[n].to(dev))
    store.refresh_bf16()
    qf.refresh()
    qf.train()
    x = torch.from_numpy(fx["x"]).to(dev).to(torch.bfloat16)
    atts = torch.from_numpy(fx["atts"]).to(dev)
    stash = {}
    out = qf.forward_hip(x, atts, stash)
    S = stash["qformer"]
    B, Q, d = x.shape[0], C["cfg"]["qf_queries"], C["cfg"]["qf_dim"]
    keys = [S["k0"]]
    for R in S["layers"]:
        keys.append(R["k1"])
        if R["cross"] is not None:
            keys.append(R["cross"]["k2"])
        keys.append(R["k3"])
    assert all(k is not None for k in keys) and len({k[2] for k in keys}) == len(keys)
    ones = torch.ones((B * Q, d), dtype=torch.bfloat16, device=dev)
    masks = [ops.dropout(ones, *k).float().cpu().view(B, Q, d) for k in keys]
    kept = torch.stack(masks).ne(0).float().mean().item()
    assert abs(kept - 0.9) < 0.01 and all(set(m.unique().tolist()) <= {0.0, float(torch.tensor(1 / 0.9).bfloat16())} for m in masks)
    masks = [m.ne(0).float() / 0.9 for m in masks]
    # attention-probability masks, one per attention call in forward order (self, cross of layer 0, self of layer 1, ...)
    H, Tk = C["cfg"]["qf_heads"], x.shape[1]
    amasks = []
    for R in S["layers"]:
        amasks.append(torch.from_numpy(G.attn_keep_mask(R["ad"][1], 0.1, B, H, Q, Q, 64, 64)) / 0.9)
        if R["cross"] is not None:
            amasks.append(torch.from_numpy(G.attn_keep_mask(R["cross"]["ad"][1], 0.1, B, H, Q, Tk, 64, 64)) / 0.9)
    assert all(abs(float(a.ne(0).float().mean()) - 0.9) < 0.03 for a in amasks)
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    ref = O.projector_qformer(Wg, C["cfg"], x.float().cpu(), atts.cpu(), hidden_masks=masks, attn_masks=amasks)
    a, g = out.float().cpu().numpy(), ref.detach().numpy()
    assert rel_err(a, g) < 3e-2 and G.cosine(g, a) > 0.9995, (rel_err(a, g), G.cosine(g, a))
    cot = torch.from_numpy(fx["cot"])
    (ref * cot.reshape(ref.shape)).sum().backward()
    qf.backward_hip(cot.to(dev).to(torch.bfloat16).reshape(-1, C["llm_dim"]).contiguous(), stash, acc=False)
    for n in W:
        gold, mine = Wg[n].grad.numpy(), store.grad_view(n).float().cpu().numpy().reshape(Wg[n].shape)
        gn = float(np.sqrt((gold.astype(np.float64) ** 2).sum()))
        if gn < 1e-4:
            continue
        assert G.cosine(gold, mine) > 0.998, (n, G.cosine(gold, mine))
        assert abs(float(np.sqrt((mine.astype(np.float64) ** 2).sum())) - gn) < 4e-2 * gn, n
    # eval mode: no dropout, no keys drawn
    qf.eval()
    calls = qf._drop_calls
    qf.forward_hip(x, atts, {})
    assert qf._drop_calls == calls


def test_cov1d_projector_matches_reference_fixture(dev):
    """f4: cov1d-linear projector forward + every parameter gradient (conv taps in the reference's [co, ci, j] layout)
    vs the fixture written by the reference's EncoderProjectorCov1d"""
    from oracle.make_golden_cases import COV1D_CASE as C
    from slam_llm_amd.model import HipProjectorCov1d, TrainableStore
    fx = G.load("cov1d")
    W = O.init_cov1d_weights(C["enc_dim"], C["llm_dim"], C["k"])
    store = TrainableStore(dev)
    pj = HipProjectorCov1d(dict(ds_rate=C["k"], enc_dim=C["enc_dim"], proj_hidden=2048, llm_dim=C["llm_dim"]), store)
    store.allocate()
    pj.bind()
    with torch.no_grad():
        for n, p in store.params.items():
            p.copy_(W[n].to(dev))
    assert set(store.params) == set(W)
    store.refresh_bf16()
    pj.refresh()
    stash = {}
    out = pj.forward_hip(torch.from_numpy(fx["x"]).to(dev).to(torch.bfloat16), stash)
    g, a = G.sub(fx, "out", out.float().cpu().numpy())
    assert rel_err(a, g) < 2e-2 and G.cosine(g, a) > 0.9998, (rel_err(a, g), G.cosine(g, a))
    cot = torch.from_numpy(fx["cot"]).to(dev).to(torch.bfloat16).reshape(-1, C["llm_dim"]).contiguous()
    for rnd, accumulate in enumerate((False, True)):   # second pass accumulates: gradients double
        if rnd:
            out = pj.forward_hip(torch.from_numpy(fx["x"]).to(dev).to(torch.bfloat16), stash)
        pj.backward_hip(cot, stash, accumulate)
        for n in W:
            mine = store.grad_view(n).float().cpu().numpy() / (rnd + 1)
            gold, sub = G.sub(fx, "grad." + n, mine)
            # the conv taps sit behind two ReLU gates evaluated on bf16-rounded pre-activations (16 rows only): a few
            # gates flip relative to fp32 -> 0.996; everything downstream of the first gate passes 0.999
            floor = 0.996 if "conv1d" in n else 0.999
            assert G.cosine(gold, sub) > floor, f"grad {n}: cosine {G.cosine(gold, sub)}"
            mn, gn = float(np.sqrt((mine.astype(np.float64) ** 2).sum())), float(fx["grad." + n + ".__norm"])
            assert abs(mn - gn) < 3e-2 * gn, f"grad {n}: norm {mn} vs {gn}"
    assert {k for k in pj.state_dict()} == {k[len("encoder_projector."):] for k in W}


def test_cov1d_projector_in_full_model_step(dev):
    """whisper -> cov1d-linear -> LLM+LoRA: loss vs the oracle path with the cov1d projector swapped in"""
    from slam_llm_amd.model import SlamHipModel
    cfg = O.make_config()
    W = O.init_weights(cfg, seed=42)
    for k_ in [k_ for k_ in W if k_.startswith("encoder_projector.")]:
        del W[k_]
    W.update(O.init_cov1d_weights(cfg["enc_dim"], cfg["llm_dim"], cfg["ds_rate"], hidden=cfg["proj_hidden"]))
    audio = O.synth_audio(2, 2.0, seed=1234)
    batch = O.synth_batch(cfg, audio, prompt_len=6, answer_lens=(5, 9), seed=1236, left_pad=True, pad_to_30s=False)
    enc = O.whisper_encoder(W, cfg, batch["audio_mel"].permute(0, 2, 1))
    proj = O.projector_cov1d(W, enc, cfg["ds_rate"])
    embeds = O.embed_splice(W["llm.base_model.model.model.embed_tokens.weight"], batch["input_ids"].clone(),
                            batch["modality_mask"].bool(), proj)
    ref_loss, _ = O.llama_forward(W, cfg, embeds, batch["attention_mask"], batch["labels"])
    model = SlamHipModel(dict(cfg, lora_dropout=0.0, projector="cov1d-linear"), dev).load_weights(W)
    model.train()
    outputs, acc = model(**{k: v.to(dev) for k, v in batch.items()})
    assert abs(float(outputs.loss.detach()) - float(ref_loss)) < 1e-2, (float(outputs.loss.detach()), float(ref_loss))
    outputs.loss.backward()
    assert float(model.store.grad_view("encoder_projector.conv1d.weight").abs().sum()) > 0


def test_c4_hubert_qformer_llm_step_matches_oracle(dev):
    """BASELINE config 4 shape (HuBERT -> Q-Former -> LLM+LoRA), tiny widths: loss vs oracle, then 2 optimizer steps"""
    from oracle.make_golden_cases import HUBERT_TINY
    from slam_llm_amd.model import SlamAdamW, SlamHipModel
    qcfg = O.qformer_config(qf_layers=2, qf_queries=8)
    cfg = dict(O.make_config(), **HUBERT_TINY, **qcfg)
    cfg.update(encoder_name="hubert", projector="q-former", enc_dim=HUBERT_TINY["hub_dim"])
    W = {k: v for k, v in O.init_weights(cfg, seed=42).items() if not k.startswith(("encoder.", "encoder_projector."))}
    W.update(O.init_hubert_weights(HUBERT_TINY, seed=7))
    W.update(O.init_qformer_weights(qcfg, cfg["enc_dim"], cfg["llm_dim"], seed=11))
    model = SlamHipModel(dict(cfg, lora_dropout=0.0, qf_dropout=0.0), dev).load_weights(W)
    model.train()
    wav = torch.nn.functional.layer_norm(O.synth_audio(2, 1.0, seed=9), (16000,))
    Q = qcfg["qf_queries"]
    samples = [O.make_sample(Q, [5, 6, 7], [9, 10, 11, 12], 2), O.make_sample(Q, [5, 6], [9, 10], 2)]
    ob = O.collate_left_pad(samples, pad_id=2)
    with torch.no_grad():
        enc = O.hubert_encoder(W, cfg, wav)
        proj = O.projector_qformer(W, qcfg, enc, None)
        emb = O.embed_splice(W["llm.base_model.model.model.embed_tokens.weight"], ob["input_ids"].clone(), ob["modality_mask"].bool(), proj)
        loss_ref, _ = O.llama_forward(W, cfg, emb, ob["attention_mask"], ob["labels"])
    gb = {k: v.to(dev) for k, v in ob.items()}
    gb["audio"] = wav.to(dev)
    opt = SlamAdamW(model, lr=1e-3)
    losses = []
    for _ in range(3):
        outputs, acc = model(**{k: v.clone() for k, v in gb.items()})
        outputs.loss.backward()
        opt.step(); opt.zero_grad()
        losses.append(float(outputs.loss.detach()))
    assert abs(losses[0] - float(loss_ref)) < 1.5e-2, (losses[0], float(loss_ref))
    assert losses[2] < losses[0]  # it trains
    assert torch.isfinite(model.store.flat).all()


def test_checkpoint_wire_format_round_trip(dev, tmp_path):
    """the trainable-only checkpoint of save_model_checkpoint_peft (utils/checkpoint_handler.py:185-201:
    {name: state_dict[name] for requires_grad params} -> model.pt) reloads through ckpt_path/load_state_dict(strict=False)
    (models/slam_model.py:44-48) and reproduces the loss bit-for-bit."""
    cfg = CASES["step_tiny"]["cfg"]
    fx = G.load("step_tiny")
    model, W = build(cfg, dev)
    model.train()
    b = batch_from_fixture(fx, dev)
    from slam_llm_amd.model import SlamAdamW
    opt = SlamAdamW(model, lr=1e-2)
    for _ in range(2):
        out, _ = model(**{k: v.clone() for k, v in b.items()})
        out.loss.backward(); opt.step(); opt.zero_grad()
    sd = model.state_dict()
    cpu_state = {}
    for name, p in model.named_parameters():      # what save_model_checkpoint_peft does
        if p.requires_grad:
            cpu_state[name] = sd[name].cpu()
    assert set(cpu_state) == set(O.trainable_names(W))
    path = tmp_path / "model.pt"
    torch.save(cpu_state, path)
    with torch.no_grad():
        ref_loss = float(model(**{k: v.clone() for k, v in b.items()})[0].loss)
    fresh, _ = build(cfg, dev)
    missing = fresh.load_state_dict(torch.load(path, map_location="cpu"), strict=False)
    assert not missing.unexpected_keys
    fresh.mark_params_updated()
    fresh.train()
    with torch.no_grad():
        got = float(fresh(**{k: v.clone() for k, v in b.items()})[0].loss)
    assert got == ref_loss


def test_lora_dropout_trains_and_is_off_in_eval(dev):
    """recipe default lora_dropout = 0.05 (asr_config.py:29-37) is live in train mode and off in eval"""
    from slam_llm_amd.model import SlamHipModel
    cfg = CASES["step_tiny"]["cfg"]
    fx = G.load("step_tiny")
    W = O.init_weights(cfg, seed=42)
    model = SlamHipModel(dict(cfg, lora_dropout=0.3), dev).load_weights(W)
    b = batch_from_fixture(fx, dev)
    model.train()
    l1 = float(model(**{k: v.clone() for k, v in b.items()})[0].loss.detach())
    out2, _ = model(**{k: v.clone() for k, v in b.items()})
    l2 = float(out2.loss.detach())
    out2.loss.backward()
    assert l1 != l2 and abs(l1 - float(fx["loss.0"])) < 0.2          # fresh mask per call, same ballpark
    assert torch.isfinite(model.store.grad).all() and float(model.store.grad.abs().sum()) > 0
    model.eval()
    with torch.no_grad():
        le = float(model(**{k: v.clone() for k, v in b.items()})[0].loss)
    assert abs(le - float(fx["loss.0"])) < 1e-2                      # dropout disabled -> reference loss


# ------------------------------------------------------------------------------------------------ generate (f1)
def _generate_setup(dev, scale):
    from oracle.make_golden_cases import GENERATE_CASE as C
    from slam_llm_amd.model import SlamHipModel
    from tests.test_oracle_golden import generate_case_weights
    fx = G.load("generate")
    W = generate_case_weights(scale)
    model = SlamHipModel(dict(C["cfg"], lora_dropout=0.0), dev).load_weights(W)
    model.eval()
    b = batch_from_fixture(fx, dev)
    return C, fx, W, model, b


@pytest.mark.parametrize("scale", [24.0, 5.0])
def test_generate_matches_reference_tokens(dev, scale):
    """SlamHipModel.generate (prefill + KV-cache decode on the HIP path, bf16) == the token ids the reference's
    slam_model.generate -> HF generate produced in fp32.  Bit-exact integer comparison; covers greedy, beam 4 / 3
    (x5 lm_head: beam search departs from greedy), eos + pad fill, length_penalty 0/1/2, left padding."""
    from tests.test_oracle_golden import GEN_RUNS, gen_key
    C, fx, W, model, b = _generate_setup(dev, scale)
    eos = int(fx[f"s{scale}.eos"])
    for nb, lp, pad, rp in GEN_RUNS:
        if rp != 1.0:
            continue   # margin-aware comparison: test_generate_with_repetition_penalty_margin_aware
        got = model.generate(**{k: v.clone() for k, v in b.items()}, max_new_tokens=C["max_new_tokens"], num_beams=nb,
                             length_penalty=lp, eos_token_id=eos, pad_token_id=pad, repetition_penalty=rp)
        want = fx[gen_key(scale, nb, lp, pad, rp)]
        assert tuple(got.shape) == want.shape and (got.cpu().numpy() == want).all(), (nb, lp, pad, got, want)


@pytest.mark.parametrize("scale", [24.0, 5.0])
def test_generate_with_repetition_penalty_margin_aware(dev, scale):
    """the two `repetition_penalty=1.3` fixture runs of the reference's generate (greedy and beam 4) on the HIP path.
    The penalty pulls an already-emitted token towards its runner-up, so some decisions have fp32 margins below the bf16
    logit noise and bit-equality with the fp32 reference is not a property of a bf16 path THERE.  Margin-aware statement:
      * noise = measured |HIP - oracle| on the prompt's next-token logits (x3 safety);
      * greedy: a row must equal the reference token for token up to (excluding) the first step whose fp32 top-2 margin is
        below 2 x noise; rows that never get that close must match completely;
      * beam 4: an item either reproduces the reference hypothesis or returns one whose fp32 score (sum of processed
        log-probs / length**length_penalty, teacher-forced through the oracle) is within the noise of the reference's."""
    from tests.test_oracle_golden import gen_key
    C, fx, W, model, b = _generate_setup(dev, scale)
    cfg, eos, L = C["cfg"], int(fx[f"s{scale}.eos"]), C["max_new_tokens"]
    cpu = {k: v.cpu() for k, v in b.items()}
    enc = O.whisper_encoder(W, cfg, cpu["audio_mel"].permute(0, 2, 1))
    emb_w = W["llm.base_model.model.model.embed_tokens.weight"]
    embeds_ref = O.embed_splice(emb_w, cpu["input_ids"].clone(), cpu["modality_mask"].bool(), O.projector_concat(W, enc, cfg["ds_rate"]))
    mask = cpu["attention_mask"].long()

    def ref_logits(toks):       # [B, t] -> fp32 next-token logits [B, V]
        x = torch.cat([embeds_ref, F.embedding(toks, emb_w)], dim=1)
        m = torch.cat([mask, torch.ones_like(toks)], dim=1)
        return O.llama_forward(W, cfg, x, m, None, position_ids=O.generate_position_ids(m))[1][:, -1, :]

    embeds, am = model(**{k: v.clone() for k, v in b.items()}, inference_mode=True)
    B, T, d = embeds.shape
    lg0, _ = model.llm.prefill(embeds.reshape(B * T, d), B, T, am, L)
    noise = 3.0 * float((lg0.float().cpu() - ref_logits(torch.zeros((B, 0), dtype=torch.int64))).abs().max())
    compared = 0
    # ---- greedy
    trace = []
    want = O.slam_generate(W, cfg, {k: v.clone() for k, v in cpu.items()}, max_new_tokens=L, num_beams=1, eos=eos, pad=1,
                           repetition_penalty=1.3, trace=trace)
    assert (want.numpy() == fx[gen_key(scale, 1, 1.0, 1, 1.3)]).all()
    got = model.generate(**{k: v.clone() for k, v in b.items()}, max_new_tokens=L, num_beams=1, eos_token_id=eos, pad_token_id=1,
                         repetition_penalty=1.3).cpu()
    margins = torch.stack([t["margin"] for t in trace])          # [steps, B]
    for r in range(B):
        close = (margins[:, r] < 2 * noise).nonzero()
        upto = int(close[0]) if close.numel() else want.shape[1]
        assert got.shape[1] >= min(upto, want.shape[1]) and torch.equal(got[r, :upto], want[r, :upto]), (scale, r, upto, got[r], want[r])
        compared += upto
    # ---- beam 4
    want = torch.from_numpy(fx[gen_key(scale, 4, 1.0, 1, 1.3)])
    got = model.generate(**{k: v.clone() for k, v in b.items()}, max_new_tokens=L, num_beams=4, length_penalty=1.0, eos_token_id=eos,
                         pad_token_id=1, repetition_penalty=1.3).cpu()

    def score(seq_rows):        # teacher-forced fp32 score of one hypothesis per item
        n = seq_rows.shape[1]
        total, length, alive = torch.zeros(B), torch.zeros(B), torch.ones(B, dtype=torch.bool)
        for t in range(n):
            lp = O._repetition_penalty(F.log_softmax(ref_logits(seq_rows[:, :t]).float(), -1), seq_rows[:, :t], 1.3)
            tok = seq_rows[:, t]
            total += torch.where(alive, lp.gather(1, tok[:, None])[:, 0], torch.zeros(B))
            length += alive.float()
            alive = alive & (tok != eos)
        return total / length.clamp(min=1)
    n = max(got.shape[1], want.shape[1])
    pad_to = lambda x: torch.cat([x, torch.full((B, n - x.shape[1]), 1, dtype=torch.int64)], 1)  # noqa: E731
    got_p, want_p = pad_to(got), pad_to(want)
    s_got, s_want = score(got_p), score(want_p)
    for r in range(B):
        if torch.equal(got_p[r], want_p[r]):
            compared += n
        else:
            assert float(s_got[r]) >= float(s_want[r]) - 2 * noise, (scale, r, float(s_got[r]), float(s_want[r]), noise, got[r], want[r])
    assert compared > 0, "nothing was comparable: the margin criterion is vacuous on this fixture"
    print(f"rp=1.3 scale {scale}: noise {noise:.4f}, token positions compared exactly: {compared}")


def test_kv_cache_decode_logits_match_oracle_teacher_forced(dev):
    """prefill + decode_step logits along the reference's greedy token sequence vs the fp32 oracle re-running the
    full sequence (no cache): bf16 bound 6e-2 + 2e-2*max|logit| (same bound as the training-forward logits test)."""
    import torch.nn.functional as F
    C, fx, W, model, b = _generate_setup(dev, 5.0)
    cfg = C["cfg"]
    toks = torch.from_numpy(fx["s5.0.tokens.b4.lp1.0.pad0"])  # [B, 12], row 0 ends with eos fill: still valid input ids
    cpu = {k: v.cpu() for k, v in b.items()}
    enc = O.whisper_encoder(W, cfg, cpu["audio_mel"].permute(0, 2, 1))
    proj = O.projector_concat(W, enc, cfg["ds_rate"])
    emb_w = W["llm.base_model.model.model.embed_tokens.weight"]
    embeds_ref = O.embed_splice(emb_w, cpu["input_ids"].clone(), cpu["modality_mask"].bool(), proj)
    mask = cpu["attention_mask"].long()
    embeds, am = model(**{k: v.clone() for k, v in b.items()}, inference_mode=True)
    B, T, d = embeds.shape
    logits, cache = model.llm.prefill(embeds.reshape(B * T, d), B, T, am, toks.shape[1])
    worst = 0.0
    for t in range(toks.shape[1]):
        x = torch.cat([embeds_ref, F.embedding(toks[:, :t], emb_w)], dim=1)
        m = torch.cat([mask, torch.ones_like(toks[:, :t])], dim=1)
        ref = O.llama_forward(W, cfg, x, m, None, position_ids=O.generate_position_ids(m))[1][:, -1, :]
        err = (logits.float().cpu() - ref).abs().max().item()
        bound = 6e-2 + 2e-2 * ref.abs().max().item()
        worst = max(worst, err / bound)
        assert err < bound, f"step {t}: logits err {err} > {bound}"
        logits = model.llm.decode_step(toks[:, t].to(dev), cache)
    print("teacher-forced decode: worst err/bound", worst)


def test_sampling_generate_on_the_hip_path(dev):
    """do_sample=True through SlamHipModel.generate (bit-exact bookkeeping vs the reference is pinned on CPU by
    tests/test_host_logic.py; a bf16 path cannot reproduce fp32 draws): degenerate settings collapse to the greedy / beam
    tokens of the reference fixture, a fixed generator reproduces itself, different seeds differ, every token respects top_k."""
    from tests.test_oracle_golden import gen_key
    C, fx, W, model, b = _generate_setup(dev, 24.0)
    eos = int(fx["s24.0.eos"])
    kw = dict(max_new_tokens=C["max_new_tokens"], eos_token_id=eos, pad_token_id=C["pad"])
    greedy = fx[gen_key(24.0, 1, 1.0, C["pad"], 1.0)]
    for deg in (dict(top_k=1), dict(top_p=1e-6, top_k=0), dict(temperature=1e-3, top_k=0)):
        got = model.generate(**{k: v.clone() for k, v in b.items()}, num_beams=1, do_sample=True, **deg, **kw)
        assert tuple(got.shape) == greedy.shape and (got.cpu().numpy() == greedy).all(), (deg, got, greedy)

    def run(seed, **extra):
        g = torch.Generator(device=dev).manual_seed(seed)
        return model.generate(**{k: v.clone() for k, v in b.items()}, do_sample=True, generator=g, **extra, **kw).cpu()
    a1, a2, a3 = run(5, num_beams=1, temperature=3.0), run(5, num_beams=1, temperature=3.0), run(6, num_beams=1, temperature=3.0)
    assert torch.equal(a1, a2) and not torch.equal(a1, a3)
    b1, b2 = run(7, num_beams=4, temperature=2.0, top_k=20), run(7, num_beams=4, temperature=2.0, top_k=20)
    assert torch.equal(b1, b2) and b1.shape[0] == a1.shape[0]
    with pytest.raises(ValueError):
        model.generate(**b, do_sample=True, temperature=0.0, eos_token_id=2, pad_token_id=0)


def test_single_utterance_inference_equals_batch_generate(dev, tmp_path):
    """model.inference(wav_path, prompt) (slam_model_asr.py:81-152) == generate() on the equivalent 1-clip batch"""
    import wave
    from types import SimpleNamespace
    from slam_llm_amd import batcher
    C, fx, W, model, b = _generate_setup(dev, 24.0)
    pcm = (O.synth_audio(1, 1.5, seed=77)[0] * 32767).round().clamp(-32768, 32767).to(torch.int16)
    path = str(tmp_path / "utt.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.numpy().tobytes())
    ids = [5, 17, 301, 42, 9]
    model.tokenizer = SimpleNamespace(encode=lambda text: list(ids), eos_token_id=int(fx["s24.0.eos"]), pad_token_id=0)
    got = model.inference(path, "transcribe", max_new_tokens=8, num_beams=4)
    audio = pcm.float() / 32768.0
    sample = batcher.make_sample(audio, ids, None, model.tokenizer.eos_token_id, batcher.whisper_audio_length(len(audio), 5))
    batch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batcher.collate([sample], 0, True).items()}
    want = model.generate(**batch, max_new_tokens=8, num_beams=4)
    assert got.shape[0] == 1 and torch.equal(got, want)


@pytest.mark.parametrize("case", ["long_sequence_single_clip", "ragged_right_padded_batch", "ragged_right_padded_batch_packed"])
def test_edge_shapes_match_oracle(dev, case):
    """edge shapes vs the fp32 oracle (loss, accuracy, every trainable gradient):
      * B = 1, 0.7 s clip (7 audio tokens), T = 1223 tokens: many attention tiles, no padding anywhere;
      * B = 3 right-padded (aispeech layout) with clips of 0.5 / 2.0 / 1.2 s and answers of 1 / 40 / 9 tokens: ragged audio
        lengths inside one zero-padded mel batch, a sample whose only label is eos, T not a multiple of 64;
      * the same batch with cfg["varlen"]: the LLM runs on the 3 sequences PACKED (no pad tokens, seg_lo / seg_hi
        attention, per-token rotary positions) -- must equal the reference's right-padded result on every valid token."""
    from slam_llm_amd.model import SlamHipModel
    cfg = O.make_config()
    W = O.init_weights(cfg, seed=42)
    if case == "long_sequence_single_clip":  # noqa: SIM102
        audio = O.synth_audio(1, 0.7, seed=21)
        batch = O.synth_batch(cfg, audio, prompt_len=900, answer_lens=(316,), seed=5, left_pad=True, pad_to_30s=False)
    else:
        g = torch.Generator().manual_seed(3)
        samples, mels = [], []
        for secs, al in ((0.5, 1), (2.0, 40), (1.2, 9)):
            a = O.synth_audio(1, secs, seed=int(secs * 100))[0]
            mel = O.log_mel_spectrogram(a[: len(a) // O.HOP * O.HOP], cfg["n_mels"]).permute(1, 0)
            mels.append(mel)
            alen = ((mel.shape[0] + 1) // 2) // cfg["ds_rate"]
            samples.append(O.make_sample(alen, torch.randint(3, cfg["vocab"], (5,), generator=g).tolist(),
                                         torch.randint(3, cfg["vocab"], (al - 1,), generator=g).tolist(), eos=2))
        batch = O.collate_right_pad(samples, pad_id=2, mels=mels)
    ref = O.train_steps({k: v.clone() for k, v in W.items()}, cfg, [{k: v.clone() for k, v in batch.items()}])[0]
    model = SlamHipModel(dict(cfg, lora_dropout=0.0, varlen=case.endswith("packed")), dev).load_weights(W)
    model.train()
    model.return_logits = True
    outputs, acc = model(**{k: v.to(dev) for k, v in batch.items()})
    outputs.loss.backward()
    assert abs(float(outputs.loss.detach()) - float(ref["loss"])) < 1e-2, (float(outputs.loss.detach()), float(ref["loss"]))
    if case.endswith("packed"):   # logits come back in the padded [B, T, V] layout: valid rows vs the oracle's
        with torch.no_grad():
            _, ref_logits, _, _ = O.slam_forward(W, cfg, {k: v.clone() for k, v in batch.items()})
        valid = batch["attention_mask"].bool()
        got = outputs.logits.float().cpu()[valid]
        want = ref_logits[valid]
        assert (got - want).abs().max() < 6e-2 + 2e-2 * want.abs().max(), float((got - want).abs().max())
        assert float(outputs.logits.float().cpu()[~valid].abs().max()) == 0.0
    n_valid = int((batch["labels"][:, 1:] != -100).sum())
    assert abs(float(acc) - float(ref["acc"])) <= 2.0 / n_valid + 1e-6
    worst = 1.0
    for n, gref in ref["grads"].items():
        mine = model.store.grad_view(n).float().cpu()
        cs = float(F.cosine_similarity(mine.flatten(), gref.flatten(), dim=0))
        worst = min(worst, cs)
        G.floor_check(cs, 0.998, f"{case}: grad {n} cosine {cs}")
        assert abs(float(mine.norm()) - float(gref.norm())) < 4e-2 * float(gref.norm()) + 1e-6, n
    print(case, "T =", batch["input_ids"].shape[1], "worst grad cosine", worst)


# ------------------------------------------------------------------------------------------------ ragged encoder (N1)
def _ragged_reference(W, cfg, audio_list, ob):
    """fp32 oracle of the ragged path: every clip through the encoder + projector ALONE (B = 1, the reference's own
    variable-length forward, encoder.py:13-30), then the usual splice / LLM / loss on the right-padded batch."""
    projs = []
    for a in audio_list:
        n = a.shape[0] // O.HOP * O.HOP
        mel = O.log_mel_spectrogram(a[:n], cfg["n_mels"]).permute(1, 0)[None]          # [1, frames, n_mels]
        enc = O.whisper_encoder(W, cfg, mel.permute(0, 2, 1))
        projs.append(O.projector_concat(W, enc, cfg["ds_rate"])[0])
    Tam = max(p.shape[0] for p in projs)
    proj = torch.stack([torch.cat([p, p.new_zeros(Tam - p.shape[0], p.shape[1])]) for p in projs])
    emb = O.embed_splice(W["llm.base_model.model.model.embed_tokens.weight"], ob["input_ids"].clone(), ob["modality_mask"].bool(), proj)
    return O.llama_forward(W, cfg, emb, ob["attention_mask"], ob["labels"])


@pytest.mark.parametrize("varlen_llm", [False, True])
def test_ragged_encoder_step_matches_per_clip_oracle(dev, varlen_llm):
    """++model_config.varlen_encoder=true, pad_or_trim off: clips of different lengths are encoded without pad frames, each
    exactly as if it were alone in the batch (stated deviation from the zero-padded reference batch).  Loss, accuracy
    and every trainable gradient vs the fp32 oracle that runs each clip through the reference's variable-length encoder with
    B = 1; with and without the packed LLM pass."""
    from slam_llm_amd import batcher
    from slam_llm_amd.model import SlamHipModel
    cfg = CASES["step_tiny"]["cfg"]
    W = O.init_weights(cfg, seed=42)
    model = SlamHipModel(dict(cfg, lora_dropout=0.0, pad_or_trim=False, varlen_encoder=True, varlen=varlen_llm), dev).load_weights(W)
    model.train()
    g = torch.Generator().manual_seed(11)
    lens = [16000 * 3 + 800, 16000 * 1 + 160 * 7, 16000 * 2, 4800]        # ragged, not multiples of the projector's 5 x 2 x 160
    audio = [(torch.randn(n, generator=g) * 0.1).clamp(-1, 1) for n in lens]
    samples = []
    for i, a in enumerate(audio):
        alen = batcher.whisper_audio_length(len(a), 5, pad_to_30s=False)
        samples.append(batcher.make_sample(a, torch.randint(3, cfg["vocab"], (4 + i,), generator=g).tolist(),
                                           torch.randint(3, cfg["vocab"], (3 + 2 * i,), generator=g).tolist(), 2, alen))
    batch = batcher.collate(samples, 0, left_pad_prompt=False)
    assert batch["audio_len_list"] == lens
    ob = {k: v for k, v in batch.items() if isinstance(v, torch.Tensor)}
    names = O.trainable_names(W)
    for n in names:
        W[n].requires_grad_(True)
    loss_ref, logits_ref = _ragged_reference(W, cfg, audio, ob)
    loss_ref.backward()
    preds = torch.argmax(logits_ref, -1)
    acc_ref = O.compute_accuracy(preds[:, :-1], ob["labels"][:, 1:], -100)
    gb = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    outputs, acc = model(**gb)
    outputs.loss.backward()
    assert abs(float(outputs.loss) - float(loss_ref)) < 1e-2, (float(outputs.loss), float(loss_ref))
    assert abs(float(acc) - float(acc_ref)) <= 1.0 / int((ob["labels"][:, 1:] != -100).sum()) + 1e-6
    for n, p in model.store.params.items():
        cs = G.cosine(W[n].grad.numpy(), p.grad.float().cpu().numpy())
        G.floor_check(cs, 0.999, f"grad {n}: cosine {cs}")


@pytest.mark.parametrize("projector,varlen_llm", [("linear", False), ("linear", True), ("cov1d-linear", True), ("q-former", False)])
def test_unfrozen_whisper_ragged_encoder_step_matches_per_clip_oracle(dev, projector, varlen_llm):
    """VERDICT r4 missing #7 (SlamHipModel used to refuse it): train_config.freeze_encoder=false TOGETHER WITH
    ++model_config.varlen_encoder=true -- the trainable Whisper encoder on the packed ragged layout (no pad frames; the conv stem's
    per-clip frame masks, packed blocks with one attention launch per clip, the window / un-window gathers in front of the stacked-row
    projectors, the padded layout in front of the Q-Former).  Loss, accuracy and EVERY gradient -- encoder, projector, LoRA -- against
    the fp32 oracle that runs each clip ALONE through the reference's variable-length encoder (B = 1) and differentiates through it.
    Floors: tests/golden_util.FLOORS["unfrozen"] / ["frozen"] as in the padded un-frozen tests."""
    from slam_llm_amd import batcher
    from slam_llm_amd.model import SlamHipModel
    cfg = dict(CASES["step_tiny"]["cfg"], lora_dropout=0.0)
    W = {k: v for k, v in O.init_weights(cfg, seed=42).items() if projector == "linear" or not k.startswith("encoder_projector.")}
    extra = {}
    if projector == "q-former":
        extra = O.qformer_config(qf_layers=2, qf_queries=8)
        W.update(O.init_qformer_weights(extra, cfg["enc_dim"], cfg["llm_dim"], seed=11))
    elif projector == "cov1d-linear":
        W.update(O.init_cov1d_weights(cfg["enc_dim"], cfg["llm_dim"], cfg["ds_rate"], hidden=cfg["proj_hidden"], seed=13))
    g = torch.Generator().manual_seed(11)
    lens = [16000 * 3 + 800, 16000 * 1 + 160 * 7, 16000 * 2, 4800]        # ragged, not multiples of the projector's 5 x 2 x 160
    audio = [(torch.randn(n, generator=g) * 0.1).clamp(-1, 1) for n in lens]
    samples = []
    for i, a in enumerate(audio):
        alen = extra["qf_queries"] if projector == "q-former" else batcher.whisper_audio_length(len(a), 5, pad_to_30s=False)
        samples.append(batcher.make_sample(a, torch.randint(3, cfg["vocab"], (4 + i,), generator=g).tolist(),
                                           torch.randint(3, cfg["vocab"], (3 + 2 * i,), generator=g).tolist(), 2, alen))
    batch = batcher.collate(samples, 0, left_pad_prompt=False, pad_or_trim=False)
    ob = {k: v for k, v in batch.items() if isinstance(v, torch.Tensor)}
    names = O.trainable_names(W) + [n for n in W if n.startswith("encoder.") and not n.endswith("positional_embedding")]
    for n in names:
        W[n].requires_grad_(True)
    projs = []
    for a in audio:       # every clip ALONE through encoder + projector (the reference's B = 1 forward), then the usual splice / LLM / loss
        mel = O.log_mel_spectrogram(a[: a.shape[0] // O.HOP * O.HOP], cfg["n_mels"]).permute(1, 0)[None]
        enc = O.whisper_encoder(W, cfg, mel.permute(0, 2, 1))
        if projector == "q-former":
            projs.append(O.projector_qformer(W, extra, enc, torch.ones(1, enc.shape[1]))[0])
        else:
            projs.append((O.projector_concat if projector == "linear" else O.projector_cov1d)(W, enc, cfg["ds_rate"])[0])
    Tam = max(p_.shape[0] for p_ in projs)
    proj = torch.stack([torch.cat([p_, p_.new_zeros(Tam - p_.shape[0], p_.shape[1])]) for p_ in projs])
    emb = O.embed_splice(W["llm.base_model.model.model.embed_tokens.weight"], ob["input_ids"].clone(), ob["modality_mask"].bool(), proj)
    loss_ref, logits_ref = O.llama_forward(W, cfg, emb, ob["attention_mask"], ob["labels"])
    loss_ref.backward()
    acc_ref = O.compute_accuracy(torch.argmax(logits_ref, -1)[:, :-1], ob["labels"][:, 1:], -100)
    grads = {n: W[n].grad.detach().clone() for n in names}
    for n in names:
        W[n].requires_grad_(False)
        W[n].grad = None
    model = SlamHipModel(dict(cfg, **extra, projector=projector, freeze_encoder=False, qf_dropout=0.0, pad_or_trim=False, varlen_encoder=True,
                              varlen=varlen_llm), dev).load_weights(W)
    model.train()
    assert set(model.store.params) == set(names)
    gb = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    outputs, acc = model(**gb)
    outputs.loss.backward()
    assert model._ragged_frames(torch.empty(4, max(lens) // 160, cfg["n_mels"]), gb) is not None      # (the ragged path really ran)
    assert abs(float(outputs.loss.detach()) - float(loss_ref.detach())) < 1.5e-2, (float(outputs.loss.detach()), float(loss_ref.detach()))
    assert abs(float(acc) - float(acc_ref)) <= 1.0 / int((ob["labels"][:, 1:] != -100).sum()) + 1e-6
    gmax = max(float(v.norm()) for v in grads.values())
    worst = (1.0, "")
    for n, p in model.store.params.items():
        gn, mine = float(grads[n].norm()), p.grad.float().cpu()
        if n.endswith("key.bias") and gn < 1e-4 * gmax:
            assert float(mine.abs().max()) < 3e-2, n
            continue
        cs = G.cosine(grads[n].numpy(), mine.numpy())
        worst = min(worst, (cs, n))
        G.floor_check(cs, G.FLOORS["unfrozen"] if (n.startswith("encoder.") or projector != "linear") else G.FLOORS["frozen"], f"grad {n}: cosine {cs}")
        assert abs(float(mine.norm()) - gn) <= 4e-2 * gn + 1e-7, f"grad {n}: norm {float(mine.norm())} vs {gn}"
    print(f"unfrozen ragged whisper + {projector} (packed LLM: {varlen_llm}): worst gradient cosine {worst}")


def test_ragged_encoder_equals_padded_path_when_nothing_is_ragged(dev):
    """equal-length batches (and B = 1): the ragged encoder equals the reference-padded path to rounding (bit for bit at B = 1 when both
    run the same attention form), and the model does not even enter it (the deviation only exists where the reference lets real frames
    attend to pad frames)"""
    from slam_llm_amd.model import SlamHipModel
    cfg = CASES["step_tiny"]["cfg"]
    W = O.init_weights(cfg, seed=42)
    m_pad = SlamHipModel(dict(cfg, lora_dropout=0.0, pad_or_trim=False), dev).load_weights(W).eval()
    m_rag = SlamHipModel(dict(cfg, lora_dropout=0.0, pad_or_trim=False, varlen_encoder=True), dev).load_weights(W).eval()
    for n_clips, n in ((1, 16000 * 2 + 480), (3, 16000)):
        audio = O.synth_audio(n_clips, n / 16000.0, seed=5)
        mel = ops_logmel(dev, audio, cfg["n_mels"], n)
        enc_pad = m_pad.encoder.forward_btc(mel)
        enc_rag, T2 = m_rag.encoder.forward_packed(mel, [n // 160] * n_clips)
        assert T2 == [enc_pad.shape[1]] * n_clips
        # same mathematics; equal to bf16 rounding: the padded path's LSE-less launches run the pre-scaled-Q form of the attention
        # forward (round 5: one more bf16 rounding of Q, accumulators started at -m), the packed rows the segment form, whose online
        # softmax also meets the keys at another tile alignment when there are several clips.  (Through round 4 the two were the same
        # kernel at B = 1 and bit-identical; with slam_attn_set_fwd_qf(60) they still are -- checked below.)
        assert rel_err(enc_rag.view(enc_pad.shape).float().cpu().numpy(), enc_pad.float().cpu().numpy()) < 1e-2
        if n_clips == 1:
            from slam_llm_amd.lib import call
            call("slam_attn_set_fwd_qf", 60)
            try:
                assert torch.equal(enc_rag.view(enc_pad.shape), m_pad.encoder.forward_btc(mel))
            finally:
                call("slam_attn_set_fwd_qf", 61)
        # the model itself takes the reference-padded path when no clip is shorter than the batch (bit-equal to the reference)
        assert m_rag._ragged_frames(mel, {"audio_len_list": [n] * n_clips}) is None
    assert m_rag._ragged_frames(mel, {"audio_len_list": [n, n - 1600, n]}) == [n // 160, (n - 1600) // 160, n // 160]


def ops_logmel(dev, audio, n_mels, n):
    from slam_llm_amd import ops
    nv = torch.full((audio.shape[0],), n, dtype=torch.int32, device=dev)
    return ops.logmel(audio.to(dev), n_mels, n_samples=n // 160 * 160, n_valid=nv, per_clip=True)


# ------------------------------------------------------------------------------------------------ ragged HuBERT (g15)
def test_hubert_ragged_batch_matches_masked_twin_fixture(dev):
    """ragged raw-audio batch with the padding mask the reference hands fairseq (slam_model.py:336): valid frames vs the fixture
    produced by the HF twin run with fairseq's frame-mask rule (bf16 path: <= 3e-2 of tensor scale)"""
    from oracle.make_golden_cases import HUBERT_TINY
    from slam_llm_amd.model import HipHubertEncoder
    fx = G.load("hubert_tiny_ragged")
    W = O.init_hubert_weights(HUBERT_TINY, seed=7)
    enc = HipHubertEncoder(dict(HUBERT_TINY), dev).load(W)
    nv = [int(x) for x in fx["n_valid"]]
    out = enc.forward_wav(torch.from_numpy(fx["wav"]).to(dev), nv)
    keep = enc.valid_frames(fx["wav"].shape[1], nv)
    pad = torch.from_numpy(fx["frame_padding_mask"])
    assert keep == (~pad).sum(1).tolist()
    got = out.float().cpu().masked_fill(pad[:, :, None], 0.0).numpy()
    g, a = G.sub(fx, "out", got)
    assert rel_err(a, g) < 3e-2 and G.cosine(g, a) > 0.9995, (rel_err(a, g), G.cosine(g, a))


def test_hubert_ragged_linear_step_matches_oracle(dev):
    """HuBERT -> linear projector -> LLM on a ragged raw-audio batch through the dataset plugin's collator (input_type raw:
    len // 320 // 5 placeholders per clip, audio_mask): loss vs the fp32 oracle, and it trains"""
    from oracle.make_golden_cases import HUBERT_TINY
    from slam_llm_amd import batcher
    from slam_llm_amd.model import SlamAdamW, SlamHipModel
    cfg = dict(O.make_config(), **HUBERT_TINY)
    cfg.update(encoder_name="hubert", enc_dim=HUBERT_TINY["hub_dim"])
    W = {k: v for k, v in O.init_weights(cfg, seed=42).items() if not k.startswith("encoder.")}
    W.update(O.init_hubert_weights(HUBERT_TINY, seed=7))
    model = SlamHipModel(dict(cfg, lora_dropout=0.0), dev).load_weights(W)
    model.train()
    g = torch.Generator().manual_seed(3)
    lens = [16000, 9000, 12345]
    samples = []
    for i, n in enumerate(lens):
        a = torch.nn.functional.layer_norm(torch.randn(n, generator=g) * 0.1, (n,))
        samples.append(batcher.make_sample(a, torch.randint(3, cfg["vocab"], (4,), generator=g).tolist(),
                                           torch.randint(3, cfg["vocab"], (3 + i,), generator=g).tolist(), 2, batcher.raw_audio_length(n)))
    batch = batcher.collate(samples, 0, left_pad_prompt=True, input_type="raw")
    assert batch["audio_mask"].sum(1).tolist() == lens and [s["audio_length"] for s in samples] == [10, 5, 7]
    ob = {k: v for k, v in batch.items() if isinstance(v, torch.Tensor)}
    with torch.no_grad():
        enc = O.hubert_encoder(W, cfg, ob["audio"], n_valid=torch.tensor(lens))
        proj = O.projector_concat(W, enc, cfg["ds_rate"])
        emb = O.embed_splice(W["llm.base_model.model.model.embed_tokens.weight"], ob["input_ids"].clone(), ob["modality_mask"].bool(), proj)
        loss_ref, _ = O.llama_forward(W, cfg, emb, ob["attention_mask"], ob["labels"])
    gb = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    opt = SlamAdamW(model, lr=1e-3)
    losses = []
    for _ in range(3):
        outputs, _ = model(**{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in gb.items()})
        outputs.loss.backward()
        opt.step(); opt.zero_grad()
        losses.append(float(outputs.loss.detach()))
    assert abs(losses[0] - float(loss_ref)) < 1.5e-2, (losses[0], float(loss_ref))
    assert losses[2] < losses[0]
