"""Generate tests/golden/*.npz by running the REFERENCE in the authoring container -- TEST INFRASTRUCTURE.

Run once here (`python oracle/make_golden.py`); /root/reference does not exist on the GPU box, so the outputs
are committed as small fixtures and every test reads only those.

What is the reference here:
  * `slam_llm.models.slam_model.slam_model.forward`   imported UNMODIFIED from /root/reference/src
  * `slam_llm.models.projector.EncoderProjectorConcat` imported UNMODIFIED
  * `slam_llm.models.encoder.WhisperWrappedEncoder.load` imported UNMODIFIED: its closure
    `extract_variable_length_features` (encoder.py:13-30) is bound onto an adapter that exposes HF
    `WhisperEncoder` submodules under openai-whisper's attribute names (openai-whisper is not installed)
  * `slam_llm.utils.metric.compute_accuracy` (via slam_model)
  * HF `LlamaForCausalLM` (transformers 5.15 here; the reference pins 4.35.2 -- loss semantics unchanged when
    num_items_in_batch is not passed), a LoRA wrapper equal to peft 0.6.0's Linear, `torch.optim.AdamW` + the
    LambdaLR of pipeline/finetune.py:253-260, loop body of utils/train_utils.py:112-169
  * HF `WhisperFeatureExtractor` numpy path for log-mel (the documented twin of whisper.log_mel_spectrogram)
Missing third-party imports (peft, soundfile, deepspeed, wandb, omegaconf, hydra, fire, whisper) are stubbed.
"""
import hashlib
import importlib.machinery
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

import transformers  # noqa: E402  (must be imported before the stubs are installed)
from transformers import LlamaConfig, LlamaForCausalLM, WhisperConfig  # noqa: E402
from transformers.models.whisper.feature_extraction_whisper import WhisperFeatureExtractor  # noqa: E402
from transformers.models.whisper.modeling_whisper import WhisperEncoder  # noqa: E402

for _n in ["soundfile", "peft", "peft.tuners", "deepspeed", "deepspeed.utils", "deepspeed.utils.zero_to_fp32",
           "wandb", "omegaconf", "hydra", "fire", "whisper"]:
    _m = MagicMock()
    _m.__spec__ = importlib.machinery.ModuleSpec(_n, None)
    sys.modules[_n] = _m
sys.path.insert(0, "/root/reference/src")
from slam_llm.models.encoder import WhisperWrappedEncoder  # noqa: E402
from slam_llm.models.projector import EncoderProjectorConcat  # noqa: E402
from slam_llm.models.slam_model import slam_model  # noqa: E402

from oracle import slam_oracle as O  # noqa: E402


class Cfg(dict):
    __getattr__ = dict.get


class _Block(nn.Module):
    def __init__(self, hf_layer):
        super().__init__()
        self.l = hf_layer

    def forward(self, x):
        out = self.l(x, attention_mask=None)
        return out[0] if isinstance(out, tuple) else out


class WhisperAdapter(nn.Module):
    """HF WhisperEncoder submodules under openai-whisper names (conv1, conv2, positional_embedding, blocks, ln_post)."""

    def __init__(self, hf: WhisperEncoder):
        super().__init__()
        self.conv1, self.conv2 = hf.conv1, hf.conv2
        self.register_buffer("positional_embedding", hf.embed_positions.weight.detach().clone())
        self.blocks = nn.ModuleList([_Block(l) for l in hf.layers])
        self.ln_post = hf.layer_norm


class LoraLinear(nn.Module):
    """peft 0.6.0 tuners/lora Linear.forward: result = base(x) + lora_B(lora_A(dropout(x))) * scaling (dropout 0)."""

    def __init__(self, base: nn.Linear, r: int, alpha: float):
        super().__init__()
        self.base = base
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, r, bias=False)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, base.out_features, bias=False)})
        self.scaling = alpha / r

    def forward(self, x):
        return self.base(x) + self.lora_B["default"](self.lora_A["default"](x)) * self.scaling


def build_reference(cfg, W):
    d = cfg["enc_dim"]
    wc = WhisperConfig(num_mel_bins=cfg["n_mels"], d_model=d, encoder_layers=cfg["enc_layers"],
                       encoder_attention_heads=cfg["enc_heads"], encoder_ffn_dim=4 * d, max_source_positions=cfg["enc_ctx"],
                       decoder_layers=1, decoder_attention_heads=cfg["enc_heads"], decoder_ffn_dim=4 * d, vocab_size=100,
                       dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, activation_function="gelu")
    wc._attn_implementation = "eager"
    hf_enc = WhisperEncoder(wc).eval()
    with torch.no_grad():
        hf_enc.conv1.weight.copy_(W["encoder.conv1.weight"]); hf_enc.conv1.bias.copy_(W["encoder.conv1.bias"])
        hf_enc.conv2.weight.copy_(W["encoder.conv2.weight"]); hf_enc.conv2.bias.copy_(W["encoder.conv2.bias"])
        hf_enc.embed_positions.weight.copy_(W["encoder.positional_embedding"])
        for i, l in enumerate(hf_enc.layers):
            p = f"encoder.blocks.{i}."
            l.self_attn.q_proj.weight.copy_(W[p + "attn.query.weight"]); l.self_attn.q_proj.bias.copy_(W[p + "attn.query.bias"])
            l.self_attn.k_proj.weight.copy_(W[p + "attn.key.weight"])
            l.self_attn.v_proj.weight.copy_(W[p + "attn.value.weight"]); l.self_attn.v_proj.bias.copy_(W[p + "attn.value.bias"])
            l.self_attn.out_proj.weight.copy_(W[p + "attn.out.weight"]); l.self_attn.out_proj.bias.copy_(W[p + "attn.out.bias"])
            l.self_attn_layer_norm.weight.copy_(W[p + "attn_ln.weight"]); l.self_attn_layer_norm.bias.copy_(W[p + "attn_ln.bias"])
            l.fc1.weight.copy_(W[p + "mlp.0.weight"]); l.fc1.bias.copy_(W[p + "mlp.0.bias"])
            l.fc2.weight.copy_(W[p + "mlp.2.weight"]); l.fc2.bias.copy_(W[p + "mlp.2.bias"])
            l.final_layer_norm.weight.copy_(W[p + "mlp_ln.weight"]); l.final_layer_norm.bias.copy_(W[p + "mlp_ln.bias"])
        hf_enc.layer_norm.weight.copy_(W["encoder.ln_post.weight"]); hf_enc.layer_norm.bias.copy_(W["encoder.ln_post.bias"])
    adapter = WhisperAdapter(hf_enc)
    # the reference's loader attaches its own extract_variable_length_features to whatever whisper.load_model returns
    sys.modules["whisper"].load_model = lambda name, device="cpu": types.SimpleNamespace(encoder=adapter)
    encoder = WhisperWrappedEncoder.load(Cfg(whisper_decode=False, encoder_path_hf=None, encoder_path="tiny"))
    assert encoder is adapter and hasattr(encoder, "extract_variable_length_features")
    for p_ in encoder.parameters():
        p_.requires_grad = False
    encoder.eval()

    lc = LlamaConfig(vocab_size=cfg["vocab"], hidden_size=cfg["llm_dim"], intermediate_size=cfg["llm_ffn"],
                     num_hidden_layers=cfg["llm_layers"], num_attention_heads=cfg["llm_heads"],
                     num_key_value_heads=cfg["llm_kv_heads"], head_dim=cfg["llm_head_dim"], rms_norm_eps=cfg["rms_eps"],
                     rope_theta=cfg["rope_theta"], max_position_embeddings=4096, attention_bias=False, mlp_bias=False,
                     tie_word_embeddings=False, attention_dropout=0.0)
    lc._attn_implementation = "eager"
    llm = LlamaForCausalLM(lc)
    P = "llm.base_model.model."
    with torch.no_grad():
        llm.model.embed_tokens.weight.copy_(W[P + "model.embed_tokens.weight"])
        for i, l in enumerate(llm.model.layers):
            p = f"{P}model.layers.{i}."
            for nme in ("q_proj", "k_proj", "v_proj", "o_proj"):
                getattr(l.self_attn, nme).weight.copy_(W[p + "self_attn." + nme + ".weight"])
            for nme in ("gate_proj", "up_proj", "down_proj"):
                getattr(l.mlp, nme).weight.copy_(W[p + "mlp." + nme + ".weight"])
            l.input_layernorm.weight.copy_(W[p + "input_layernorm.weight"])
            l.post_attention_layernorm.weight.copy_(W[p + "post_attention_layernorm.weight"])
        llm.model.norm.weight.copy_(W[P + "model.norm.weight"])
        llm.lm_head.weight.copy_(W[P + "lm_head.weight"])
    for p_ in llm.parameters():
        p_.requires_grad = False
    llm.eval()
    for i, l in enumerate(llm.model.layers):  # get_peft_model(LoraConfig(target_modules=...)) equivalent
        p = f"{P}model.layers.{i}.self_attn."
        for nme in cfg["lora_targets"]:
            ll = LoraLinear(getattr(l.self_attn, nme), cfg["lora_r"], cfg["lora_alpha"])
            with torch.no_grad():
                ll.lora_A["default"].weight.copy_(W[p + nme + ".lora_A.default.weight"])
                ll.lora_B["default"].weight.copy_(W[p + nme + ".lora_B.default.weight"])
            setattr(l.self_attn, nme, ll)

    mc = Cfg(encoder_name="whisper", encoder_projector="linear", encoder_projector_ds_rate=cfg["ds_rate"],
             encoder_dim=cfg["enc_dim"], llm_dim=cfg["llm_dim"])
    proj = EncoderProjectorConcat(mc)
    with torch.no_grad():
        proj.linear1.weight.copy_(W["encoder_projector.linear1.weight"]); proj.linear1.bias.copy_(W["encoder_projector.linear1.bias"])
        proj.linear2.weight.copy_(W["encoder_projector.linear2.weight"]); proj.linear2.bias.copy_(W["encoder_projector.linear2.bias"])
    tc = Cfg(freeze_encoder=True, enable_deepspeed=False)
    model = slam_model(encoder, llm, proj, None, tc, mc, metric="acc")
    model.train()
    return model


def ref_trainables(model, cfg):
    """name -> parameter, in the build's naming (peft-style keys)."""
    out = {}
    for n in ("linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias"):
        out["encoder_projector." + n] = model.encoder_projector.get_parameter(n)
    for i, l in enumerate(model.llm.model.layers):
        for nme in cfg["lora_targets"]:
            m = getattr(l.self_attn, nme)
            p = f"llm.base_model.model.model.layers.{i}.self_attn.{nme}."
            out[p + "lora_A.default.weight"] = m.lora_A["default"].weight
            out[p + "lora_B.default.weight"] = m.lora_B["default"].weight
    return out


def pack(fx, name, arr, limit=8192):
    """big tensors are stored as a strided subsample + L2 norm + sum (keeps the fixtures small)"""
    a = np.asarray(arr, dtype=np.float32).reshape(-1)
    stride = max(1, -(-a.size // limit))
    fx[name] = a[::stride].copy()
    fx[name + ".__stride"] = np.int64(stride)
    fx[name + ".__norm"] = np.float64(np.sqrt((a.astype(np.float64) ** 2).sum()))
    fx[name + ".__sum"] = np.float64(a.astype(np.float64).sum())


def wsum(W):
    h = hashlib.sha256()
    for k in sorted(W):
        h.update(k.encode()); h.update(W[k].detach().numpy().tobytes())
    return h.hexdigest()


def gen_mel():
    """log-mel fixtures from HF WhisperFeatureExtractor (numpy path) for 80 and 128 mel bins."""
    audio = O.synth_audio(2, 3.7, seed=1234)
    out = {"audio": audio.numpy()}
    for nm in (80, 128):
        fe = WhisperFeatureExtractor(feature_size=nm, sampling_rate=16000, hop_length=160, chunk_length=30, n_fft=400)
        feats = fe([a.numpy() for a in audio], sampling_rate=16000, return_tensors="np", padding="max_length")["input_features"]
        feats = np.asarray(feats, dtype=np.float32)  # [2, nm, 3000]
        idx = np.unique(np.concatenate([np.arange(0, 3000, 7), np.arange(0, 24), np.arange(360, 380), np.arange(2976, 3000)]))
        out[f"mel{nm}_frames"] = idx.astype(np.int32)
        out[f"mel{nm}_values"] = feats[:, :, idx]
        out[f"mel{nm}_max"] = feats.reshape(2, -1).max(axis=1)
    np.savez_compressed(os.path.join(GOLD, "logmel.npz"), **out)
    print("logmel.npz written")


def gen_step(name, cfg, clip_seconds, answer_lens, n_steps=3, left_pad=True):
    torch.manual_seed(0)
    W = O.init_weights(cfg, seed=42)
    model = build_reference(cfg, W)
    audio = O.synth_audio(len(answer_lens), clip_seconds, seed=1234)
    batch = O.synth_batch(cfg, audio, prompt_len=6, answer_lens=answer_lens, seed=1236, left_pad=left_pad, pad_to_30s=False)
    tr = ref_trainables(model, cfg)
    for p_ in tr.values():
        p_.requires_grad = True
    opt = torch.optim.AdamW(list(tr.values()), lr=1e-2, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: O.lr_lambda(s, 2, 10))
    fx = {"weights_sha256": np.array(wsum(W)), "audio": audio.numpy()}
    for k, v in batch.items():
        fx["batch." + k] = v.numpy()
    for step in range(n_steps):
        b = {k: v.clone() for k, v in batch.items()}
        outputs, acc = model(**b)  # the reference forward, unmodified
        loss = outputs.loss
        loss.backward()
        if step == 0:
            pack(fx, "logits", outputs.logits.detach().float().numpy(), limit=65536)
            fx["logits.shape"] = np.array(outputs.logits.shape)
            with torch.no_grad():
                enc = model.encoder.extract_variable_length_features(batch["audio_mel"].permute(0, 2, 1))
                pack(fx, "encoder_out", enc.numpy(), limit=32768)
                pack(fx, "projector_out", model.encoder_projector(enc).numpy(), limit=32768)
            for n, p_ in tr.items():
                pack(fx, "grad." + n, p_.grad.detach().numpy())
        fx[f"loss.{step}"] = np.float32(loss.item())
        fx[f"acc.{step}"] = np.float32(float(acc))
        opt.step(); sched.step(); opt.zero_grad()
        print(f"{name} step {step}: loss {loss.item():.6f} acc {float(acc):.4f}")
    for n, p_ in tr.items():
        pack(fx, "final." + n, p_.detach().numpy())
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **fx)
    print(name + ".npz written")


from oracle.make_golden_cases import CASES, COV1D_CASE, GENERATE_CASE, HUBERT_TINY, QFORMER_CASE  # noqa: E402

def gen_cov1d():
    """EncoderProjectorCov1d (models/projector.py:29-49) imported UNMODIFIED: forward + every parameter gradient."""
    from slam_llm.models.projector import EncoderProjectorCov1d
    c = COV1D_CASE
    W = O.init_cov1d_weights(c["enc_dim"], c["llm_dim"], c["k"])
    m = EncoderProjectorCov1d(Cfg(encoder_projector_ds_rate=c["k"], encoder_dim=c["enc_dim"], llm_dim=c["llm_dim"]))
    m.load_state_dict({k[len("encoder_projector."):]: v for k, v in W.items()})
    g = torch.Generator().manual_seed(6)
    x = torch.randn(c["B"], c["T"], c["enc_dim"], generator=g)
    cot = torch.randn(c["B"], c["T"] // c["k"], c["llm_dim"], generator=g)
    out = m(x)
    (out * cot).sum().backward()
    fx = {"x": x.numpy(), "cot": cot.numpy(), "weights_sha256": np.array(wsum(W))}
    pack(fx, "out", out.detach().numpy(), limit=65536)
    for n, p_ in m.named_parameters():
        pack(fx, "grad.encoder_projector." + n, p_.grad.numpy(), limit=8192)
    np.savez_compressed(os.path.join(GOLD, "cov1d.npz"), **fx)
    print("cov1d.npz written", tuple(out.shape))


def gen_generate():
    """slam_model.generate (slam_model.py:409-456) UNMODIFIED -> HF LlamaForCausalLM.generate (fp32, transformers
    5.15 here), greedy and beam-4, on a ragged left-padded inference batch."""
    case = GENERATE_CASE
    cfg = case["cfg"]
    audio = O.synth_audio(len(case["clip_samples"]), 2.0, seed=1234)
    batch = O.synth_infer_batch(cfg, audio, case["clip_samples"], case["prompt_lens"])
    fx = {"audio": audio.numpy()}
    for k, v in batch.items():
        fx["batch." + k] = v.numpy()
    for scale in case["lm_head_scales"]:
        W = O.init_weights(cfg, seed=42)
        W["llm.base_model.model.lm_head.weight"] = W["llm.base_model.model.lm_head.weight"] * scale
        model = build_reference(cfg, W)
        model.eval()
        fx[f"s{scale}.weights_sha256"] = np.array(wsum(W))

        def run(eos, nb, **kw):
            model.tokenizer = types.SimpleNamespace(bos_token_id=case["bos"], eos_token_id=eos, pad_token_id=kw.pop("pad"))
            b = {k: v.clone() for k, v in batch.items()}
            with torch.no_grad():
                return model.generate(**b, max_new_tokens=case["max_new_tokens"], num_beams=nb, **kw)

        # pick eos: a token some greedy row emits mid-sequence, so rows (and beams) finish at different lengths
        free = run(cfg["vocab"] - 1, 1, pad=case["pad"])
        eos = int(free[0, 4])
        fx[f"s{scale}.eos"] = np.int64(eos)
        for nb, lp, pad, rp in ((1, 1.0, case["pad"], 1.0), (4, 1.0, case["pad"], 1.0), (4, 2.0, 1, 1.0), (3, 0.0, 1, 1.0),
                                (1, 1.0, 1, 1.3), (4, 1.0, 1, 1.3)):
            out = run(eos, nb, length_penalty=lp, pad=pad, repetition_penalty=rp)
            mine = O.slam_generate(W, cfg, {k: v.clone() for k, v in batch.items()}, max_new_tokens=case["max_new_tokens"],
                                   num_beams=nb, length_penalty=lp, eos=eos, pad=pad, repetition_penalty=rp)
            ok = out.shape == mine.shape and bool((out == mine).all())
            print(f"generate scale={scale} beams={nb} lp={lp} pad={pad} rp={rp}: oracle match {ok}\n{out.numpy()}")
            if not ok:
                print("oracle:\n", mine.numpy())
            fx[f"s{scale}.tokens.b{nb}.lp{lp}.pad{pad}" + (f".rp{rp}" if rp != 1.0 else "")] = out.numpy()
    np.savez_compressed(os.path.join(GOLD, "generate.npz"), **fx)
    print("generate.npz written")


def gen_batcher():
    """Run the reference's own window_class + MultiTaskDynamicBatchDataset (speech_dataset_large.py:235-263).
    The module itself cannot be imported (kaldiio/whisper missing), so the two definitions are exec'd from its
    source text, unmodified."""
    import ast
    from functools import partial
    src = open("/root/reference/src/slam_llm/datasets/speech_dataset_large.py").read()
    tree = ast.parse(src)
    ns = {"IterableDataset": torch.utils.data.IterableDataset}
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in ("MultiTaskDynamicBatchDataset", "window_class"):
            exec(compile(ast.Module([node], []), "speech_dataset_large.py", "exec"), ns)

    class Fake(torch.utils.data.IterableDataset):
        def __init__(self, lens):
            self.lens = lens
            self.collator = None

        def __iter__(self):
            for i, n in enumerate(self.lens):
                yield {"idx": i, "input_ids": torch.zeros(n)}

    rng = np.random.RandomState(1235)
    fx = {}
    for ci, (lens, mfl) in enumerate([
        ([380, 380, 100, 500, 90, 90, 90, 700, 10], 1000),
        ([380] * 64, 12000),
        (list(rng.randint(40, 400, size=300)), 12000),
        (list(rng.randint(40, 400, size=50)), 300),
    ]):
        ds = ns["MultiTaskDynamicBatchDataset"](Fake(lens), partial(ns["window_class"], max_frame_length=mfl))
        groups = [[e["idx"] for e in g] for g in ds]
        fx[f"lens.{ci}"] = np.array(lens, dtype=np.int64)
        fx[f"mfl.{ci}"] = np.int64(mfl)
        fx[f"group_sizes.{ci}"] = np.array([len(g) for g in groups], dtype=np.int64)
        assert sum(groups, []) == list(range(len(lens)))
    np.savez_compressed(os.path.join(GOLD, "batcher.npz"), **fx)
    print("batcher.npz written")


def gen_hubert():
    """HF HubertModel (the runnable twin of fairseq's HuBERT, which is not installed) on a tiny config."""
    from transformers import HubertConfig, HubertModel
    cfg = HUBERT_TINY
    hc = HubertConfig(hidden_size=cfg["hub_dim"], num_hidden_layers=cfg["hub_layers"], num_attention_heads=cfg["hub_heads"],
                      intermediate_size=cfg["hub_ffn"], conv_dim=list(cfg["hub_conv_dim"]), conv_kernel=list(cfg["hub_conv_kernel"]),
                      conv_stride=list(cfg["hub_conv_stride"]), conv_bias=True, feat_extract_norm="layer",
                      do_stable_layer_norm=True, feat_proj_layer_norm=True, num_conv_pos_embeddings=cfg["hub_pos_k"],
                      num_conv_pos_embedding_groups=cfg["hub_pos_groups"], hidden_dropout=0.0, attention_dropout=0.0,
                      activation_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0, mask_feature_prob=0.0,
                      layer_norm_eps=cfg["hub_eps"], hidden_act="gelu", feat_extract_activation="gelu")
    hc._attn_implementation = "eager"
    m = HubertModel(hc).eval()
    W = O.init_hubert_weights(cfg, seed=7)
    sd = m.state_dict()
    with torch.no_grad():
        for k in sd:
            if "pos_conv_embed.conv.parametrizations" in k or k == "masked_spec_embed":
                continue
            sd[k].copy_(W["encoder." + k])
        # weight-norm parametrisation (dim=2): set v = w and g = ||w|| so that the effective weight equals ours
        w = W["encoder.encoder.pos_conv_embed.conv.weight"]
        sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"].copy_(w)
        sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"].copy_(w.norm(dim=(0, 1), keepdim=True))
    m.load_state_dict(sd)
    wav = O.synth_audio(2, 1.0, seed=4321)
    wav = torch.nn.functional.layer_norm(wav, (wav.shape[1],))  # dataset `normalize` (speech_dataset.py:96-97)
    with torch.no_grad():
        out = m(wav).last_hidden_state
    fx = {"wav": wav.numpy(), "weights_sha256": np.array(wsum(W)), "out_shape": np.array(out.shape)}
    pack(fx, "out", out.numpy(), limit=65536)
    np.savez_compressed(os.path.join(GOLD, "hubert_tiny.npz"), **fx)
    print("hubert_tiny.npz written", tuple(out.shape))


def gen_hubert_ragged():
    """Ragged raw-audio batch through the HF twin of fairseq's HuBERT with a padding mask.  The frame-level mask is fairseq's
    rule (the reference calls fairseq, slam_model.py:336; HF derives frame lengths from the conv arithmetic instead and can
    differ by one frame), injected by overriding HF's `_get_feature_vector_attention_mask`; everything downstream (padded frames
    zeroed before the positional conv, masked as keys in every layer) is HF's own code, identical to fairseq's."""
    from transformers import HubertConfig, HubertModel
    cfg = HUBERT_TINY
    hc = HubertConfig(hidden_size=cfg["hub_dim"], num_hidden_layers=cfg["hub_layers"], num_attention_heads=cfg["hub_heads"],
                      intermediate_size=cfg["hub_ffn"], conv_dim=list(cfg["hub_conv_dim"]), conv_kernel=list(cfg["hub_conv_kernel"]),
                      conv_stride=list(cfg["hub_conv_stride"]), conv_bias=True, feat_extract_norm="layer",
                      do_stable_layer_norm=True, feat_proj_layer_norm=True, num_conv_pos_embeddings=cfg["hub_pos_k"],
                      num_conv_pos_embedding_groups=cfg["hub_pos_groups"], hidden_dropout=0.0, attention_dropout=0.0,
                      activation_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0, mask_feature_prob=0.0,
                      layer_norm_eps=cfg["hub_eps"], hidden_act="gelu", feat_extract_activation="gelu")
    hc._attn_implementation = "eager"
    m = HubertModel(hc).eval()
    W = O.init_hubert_weights(cfg, seed=7)
    sd = m.state_dict()
    with torch.no_grad():
        for k in sd:
            if "pos_conv_embed.conv.parametrizations" in k or k == "masked_spec_embed":
                continue
            sd[k].copy_(W["encoder." + k])
        w = W["encoder.encoder.pos_conv_embed.conv.weight"]
        sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"].copy_(w)
        sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"].copy_(w.norm(dim=(0, 1), keepdim=True))
    m.load_state_dict(sd)
    n_valid = torch.tensor([16000, 9000, 12345])
    N = int(n_valid.max())
    clips = O.synth_audio(3, 1.0, seed=4322)
    wav = torch.zeros(3, N)
    for b, n in enumerate(n_valid.tolist()):   # dataset `normalize` per clip (speech_dataset.py:96-97), collator zero padding (:238-244)
        wav[b, :n] = torch.nn.functional.layer_norm(clips[b, :n], (n,))
    amask = (torch.arange(N)[None, :] < n_valid[:, None]).long()

    def fairseq_frame_mask(feature_vector_length, attention_mask):
        return ~O.hubert_frame_padding_mask(attention_mask.shape[1], feature_vector_length, attention_mask.sum(-1))
    m._get_feature_vector_attention_mask = fairseq_frame_mask
    with torch.no_grad():
        out = m(wav, attention_mask=amask).last_hidden_state
    pad = O.hubert_frame_padding_mask(N, out.shape[1], n_valid)
    fx = {"wav": wav.numpy(), "n_valid": n_valid.numpy(), "weights_sha256": np.array(wsum(W)), "out_shape": np.array(out.shape),
          "frame_padding_mask": pad.numpy()}
    valid = out.masked_fill(pad[:, :, None], 0.0)     # rows of padded frames are unspecified: compare valid frames only
    pack(fx, "out", valid.numpy(), limit=65536)
    np.savez_compressed(os.path.join(GOLD, "hubert_tiny_ragged.npz"), **fx)
    print("hubert_tiny_ragged.npz written", tuple(out.shape), "valid frames", (~pad).sum(1).tolist())


def gen_qformer():
    """The reference's EncoderProjectorQFormer (projector.py:51-80), imported unmodified, in eval mode (dropout off):
    output and gradients of every parameter for a fixed random cotangent."""
    from slam_llm.models.projector import EncoderProjectorQFormer
    c = QFORMER_CASE
    cfg = c["cfg"]
    mc = Cfg(encoder_dim=c["enc_dim"], llm_dim=c["llm_dim"], qformer_layers=cfg["qf_layers"], query_len=cfg["qf_queries"])
    torch.manual_seed(0)
    m = EncoderProjectorQFormer(mc).eval()
    W = O.init_qformer_weights(cfg, c["enc_dim"], c["llm_dim"], seed=11)
    sd = m.state_dict()
    assert set("encoder_projector." + k for k in sd) == set(W), set("encoder_projector." + k for k in sd) ^ set(W)
    m.load_state_dict({k: W["encoder_projector." + k] for k in sd})
    g = torch.Generator().manual_seed(5)
    x = torch.randn(c["B"], c["Tk"], c["enc_dim"], generator=g)
    atts = torch.ones(c["B"], c["Tk"])
    atts[1, -c["masked_tail"]:] = 0
    cot = torch.randn(c["B"], cfg["qf_queries"], c["llm_dim"], generator=g)
    out = m(x, atts)
    (out * cot).sum().backward()
    fx = {"x": x.numpy(), "atts": atts.numpy(), "cot": cot.numpy(), "weights_sha256": np.array(wsum(W))}
    pack(fx, "out", out.detach().numpy(), limit=65536)
    for n, p_ in m.named_parameters():
        pack(fx, "grad.encoder_projector." + n, p_.grad.numpy(), limit=4096)
    np.savez_compressed(os.path.join(GOLD, "qformer.npz"), **fx)
    print("qformer.npz written", tuple(out.shape))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 1:  # regenerate selected fixtures only: python oracle/make_golden.py generate qformer
        for nme in sys.argv[1:]:
            globals()["gen_" + nme]()
        sys.exit(0)
    gen_generate()
    gen_cov1d()
    gen_mel()
    gen_batcher()
    gen_hubert()
    gen_hubert_ragged()
    gen_qformer()
    for nme, c in CASES.items():
        gen_step(nme, **c)
