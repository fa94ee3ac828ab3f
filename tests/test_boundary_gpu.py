"""GPU: the plugin boundary end to end (model_factory -> train step -> inference_batch), fresh-module initialisation of
the trainables, optimizer state round trip, and ONE true-width parity case (Whisper-large-v3 / Llama-3-8B layer widths,
V = 128 256) against the fp32 oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import slam_oracle as O
from tests import golden_util as G
from tests.test_plugin_boundary import Cfg, MergingTok, PLUGIN, loaders, recipe_configs

pytestmark = pytest.mark.gpu

TINY_ARCH = dict(enc_dim=128, enc_heads=2, enc_layers=2, n_mels=80, llm_dim=128, llm_layers=2, llm_heads=2, llm_kv_heads=1,
                 llm_head_dim=64, llm_ffn=256, vocab=512)


def _tiny_recipe(peft=None, **train_over):
    return recipe_configs("asr_librispeech", peft=peft,
                          model=dict(file=PLUGIN + ":model_factory", encoder_name="whisper", encoder_path="/ckpt/whisper/tiny.pt",
                                     llm_name="tinyllama-1.1b", encoder_dim=128, llm_dim=128, arch_overrides=TINY_ARCH, random_init=True),
                          train=dict(use_peft=True, freeze_encoder=True, freeze_llm=True, **train_over))


def test_model_factory_train_step_and_inference_batch(dev, tmp_path):
    """the plugin resolved by the reference's loader, built from the reference's recipe config defaults (+ overrides), one
    optimizer step with torch.optim.AdamW(model.parameters()) as finetune.py:247-251 builds it, then the batch-decode loop
    of pipeline/inference_batch.py:118-137 writing <decode_log>_pred / _gt."""
    get_factory, _, _ = loaders()
    import logging
    from slam_llm_amd.slam_model_hip import inference_batch
    tc, mc, _ = _tiny_recipe()
    factory = get_factory(mc, logging.getLogger("t"))
    model, tokenizer = factory(tc, mc, metric="acc")
    assert tokenizer is None and model.autograd_params is False            # enable_ddp defaults to False in the recipes
    cfg = model.cfg
    assert cfg["lora_r"] == 8 and cfg["lora_targets"] == ("q_proj", "v_proj") and cfg["lora_dropout"] == 0.05   # asr_config.py:29-37
    model = model.cuda(0)                                                   # finetune.py:181 must be a no-op move
    with pytest.raises(RuntimeError, match="fp32 trainable masters"):
        model.to(torch.float16)                                             # (fp16 masters are refused loudly; the pure_bf16 route --
    #                                                                         model.to(torch.bfloat16), finetune.py:154-155 -- is accepted: tests/test_amp_rccl_gpu.py)
    model.train()
    audio = O.synth_audio(2, 1.0, seed=5)
    ob = O.synth_batch(cfg, audio, prompt_len=4, answer_lens=(3, 6), seed=6, left_pad=True, pad_to_30s=False)
    gb = {k: v.to(dev) for k, v in ob.items()}
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=0.0)
    losses = []
    for _ in range(3):
        outputs, acc = model(**{k: v.clone() for k, v in gb.items()})
        outputs.loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(float(outputs.loss))
    assert losses[2] < losses[0] and np.isfinite(losses).all()
    # batch decode
    model.eval()
    model.tokenizer = MergingTok()
    ib = O.synth_infer_batch(cfg, O.synth_audio(3, 2.0, seed=8), clip_samples=(32000, 22400, 28800), prompt_lens=(6, 4, 7))
    ib = dict(ib, keys=["a", "b", "c"], targets=["ref a", "ref b", "ref c"])
    n = inference_batch(model, model.tokenizer, [ib], str(tmp_path / "decode"), device=dev, max_new_tokens=5, num_beams=2)
    pred = (tmp_path / "decode_pred").read_text().splitlines()
    gt = (tmp_path / "decode_gt").read_text().splitlines()
    assert n == 3 and [l.split("\t")[0] for l in pred] == ["a", "b", "c"] and gt == ["a\tref a", "b\tref b", "c\tref c"]


def test_enable_ddp_recipe_flag_selects_autograd_params(dev):
    get_factory, _, _ = loaders()
    import logging
    tc, mc, _ = _tiny_recipe(peft=dict(lora_dropout=0.0), enable_ddp=True)   # dropout off: two backward passes must be identical
    model, _ = get_factory(mc, logging.getLogger("t"))(tc, mc)
    assert model.autograd_params is True
    model.train()
    cfg = model.cfg
    ob = O.synth_batch(cfg, O.synth_audio(2, 1.0, seed=5), prompt_len=4, answer_lens=(3, 6), seed=6, left_pad=True, pad_to_30s=False)
    gb = {k: v.to(dev) for k, v in ob.items()}
    out, _ = model(**{k: v.clone() for k, v in gb.items()})
    out.loss.backward()
    g1 = {n: p.grad.clone() for n, p in model.named_parameters()}
    assert all(g is not None and torch.isfinite(g).all() for g in g1.values())
    out, _ = model(**{k: v.clone() for k, v in gb.items()})
    out.loss.backward()                                    # autograd accumulates into .grad (AccumulateGrad), like any module
    for n, p in model.named_parameters():
        assert torch.allclose(p.grad, 2 * g1[n], rtol=1e-5, atol=1e-8), n


def test_fresh_finetune_initialises_projector_and_lora_like_the_reference(dev, tmp_path):
    """model_factory on pretrained encoder/LLM state dicts WITHOUT a projector/LoRA checkpoint used to leave
    the trainables at zero (a fixed point).  They must be initialised like the reference's fresh modules: nn.Linear
    reset_parameters bounds for the projector (projector.py:11-13), peft's kaiming-uniform A / zero B -- and train."""
    from slam_llm_amd.slam_model_hip import model_factory
    cfg = O.make_config()
    W = O.init_weights(cfg, seed=42)
    frozen = {k: v for k, v in W.items() if k not in O.trainable_names(W)}
    torch.save({k: v for k, v in frozen.items() if k.startswith("encoder.")}, tmp_path / "enc.pt")
    torch.save({k: v for k, v in frozen.items() if not k.startswith("encoder.")}, tmp_path / "llm.pt")
    tc, mc, _ = recipe_configs("asr_librispeech",
                               model=dict(encoder_name="whisper", encoder_path="/x/tiny.pt", llm_name="tinyllama-1.1b", encoder_dim=128, llm_dim=128,
                                          arch_overrides=TINY_ARCH, encoder_state=str(tmp_path / "enc.pt"), llm_state=str(tmp_path / "llm.pt")),
                               train=dict(use_peft=True, freeze_encoder=True, freeze_llm=True, seed=1234))
    model, _ = model_factory(tc, mc)
    model2, _ = model_factory(tc, mc)
    st = model.store
    assert torch.equal(st.flat, model2.store.flat), "seeded init must agree across ranks"
    w1 = st.params["encoder_projector.linear1.weight"]
    bound = 1.0 / np.sqrt(w1.shape[1])
    assert float(w1.abs().max()) <= bound * (1 + 1e-6) and abs(float(w1.std()) - bound / np.sqrt(3)) < 0.05 * bound
    b1 = st.params["encoder_projector.linear1.bias"]
    assert 0 < float(b1.abs().max()) <= bound * (1 + 1e-6)
    for n, p in st.params.items():
        if "lora_B" in n:
            assert float(p.abs().max()) == 0.0, n
        elif "lora_A" in n:
            assert 0 < float(p.abs().max()) <= 1.0 / np.sqrt(p.shape[1]) * (1 + 1e-6), n
    model.train()
    ob = O.synth_batch(cfg, O.synth_audio(2, 1.0, seed=5), prompt_len=4, answer_lens=(3, 6), seed=6, left_pad=True, pad_to_30s=False)
    gb = {k: v.to(dev) for k, v in ob.items()}
    out, _ = model(**{k: v.clone() for k, v in gb.items()})
    out.loss.backward()
    assert float(st.grad_view("encoder_projector.linear1.weight").abs().max()) > 0
    assert float(st.grad_view("encoder_projector.linear2.weight").abs().max()) > 0
    nz_b = [float(st.grad_view(n).abs().max()) for n in st.params if "lora_B" in n]
    assert min(nz_b) > 0, "dB must be non-zero when A is initialised (B = 0 only zeroes dA on the first step)"


def test_raw_wavlm_checkpoint_supplies_its_regularisers_and_trains(dev, tmp_path):
    """models/encoder.py:118-121 reads a WavLM checkpoint as {"cfg": ..., "model": state_dict} and builds the module from that cfg.  Given
    as `encoder_state` with train_config.freeze_encoder=false, the factory unwraps it (keys gain the `encoder.model.` prefix), takes
    dropout / attention_dropout / activation_dropout / dropout_input / encoder_layerdrop from ITS cfg (a model_config.encoder_* value
    would win), and the un-frozen encoder then runs in train mode with them: masks are drawn (the counter moves), eval mode draws none,
    every encoder parameter outside a skipped layer gets a gradient."""
    from oracle.make_golden_cases import WAVLM_TINY
    from slam_llm_amd.slam_model_hip import model_factory
    cfg = dict(O.make_config(), **WAVLM_TINY)
    cfg.update(encoder_name="wavlm", enc_dim=WAVLM_TINY["hub_dim"])
    W = O.init_weights(cfg, seed=42)
    enc_sd = {k[len("encoder.model."):]: v for k, v in O.init_wavlm_weights(WAVLM_TINY, seed=9).items()}
    torch.save({"cfg": {"dropout": 0.2, "attention_dropout": 0.05, "activation_dropout": 0.0, "encoder_layerdrop": 0.0, "dropout_input": 0.1,
                        "encoder_layers": WAVLM_TINY["hub_layers"]}, "model": enc_sd}, tmp_path / "WavLM-Tiny.pt")
    torch.save({k: v for k, v in W.items() if k.startswith("llm.") and k not in O.trainable_names(W)}, tmp_path / "llm.pt")
    arch = dict(TINY_ARCH, **{k: (list(v) if isinstance(v, tuple) else v) for k, v in WAVLM_TINY.items()}, enc_dim=WAVLM_TINY["hub_dim"])
    tc, mc, _ = recipe_configs("asr_librispeech",
                               model=dict(encoder_name="wavlm", encoder_path="/ckpt/WavLM-Large.pt", llm_name="tinyllama-1.1b",
                                          encoder_dim=WAVLM_TINY["hub_dim"], llm_dim=128, arch_overrides=arch, normalize=True,
                                          encoder_state=str(tmp_path / "WavLM-Tiny.pt"), llm_state=str(tmp_path / "llm.pt"),
                                          encoder_attention_dropout=0.15),
                               train=dict(use_peft=True, freeze_encoder=False, freeze_llm=True, seed=7))
    model, _ = model_factory(tc, mc)
    c = model.encoder.cfg
    assert (c["hub_dropout"], c["hub_dropout_input"], c["hub_layerdrop"], c["hub_activation_dropout"]) == (0.2, 0.1, 0.0, 0.0)
    assert c["hub_attention_dropout"] == 0.15                      # the recipe's own value wins over the checkpoint's 0.05
    assert "encoder.model.encoder.layers.0.self_attn.grep_a" in model.store.params
    got = model.store.params["encoder.model.post_extract_proj.weight"].detach().cpu()
    assert torch.equal(got, enc_sd["post_extract_proj.weight"])
    model.train()
    N = 16000
    wav = torch.nn.functional.layer_norm(O.synth_audio(2, 1.0, seed=9), (N,))
    alen = N // 320 // 5
    ob = O.collate_left_pad([O.make_sample(alen, [5, 6, 7], [9, 10, 11], 2), O.make_sample(alen, [5, 6], [9, 10], 2)], pad_id=2)
    gb = {k: v.to(dev) for k, v in ob.items()}
    gb["audio"] = wav.to(dev)
    gb["audio_len"] = torch.tensor([N, N], dtype=torch.int32, device=dev)
    calls = model.encoder._drop_calls
    out, _ = model(**{k: v.clone() for k, v in gb.items()})
    out.loss.backward()
    assert model.encoder._drop_calls > calls                       # train mode: masks were drawn
    assert float(model.store.grad_view("encoder.model.encoder.layers.1.fc1.weight").abs().max()) > 0
    assert float(model.store.grad_view("encoder.model.feature_extractor.conv_layers.0.0.weight").abs().max()) > 0
    l_train = float(out.loss.detach())
    model.eval()
    calls = model.encoder._drop_calls
    with torch.no_grad():
        e1, _ = model(**{k: v.clone() for k, v in gb.items()})
        e2, _ = model(**{k: v.clone() for k, v in gb.items()})
    assert model.encoder._drop_calls == calls and float(e1.loss) == float(e2.loss) and np.isfinite(l_train)


def test_slam_adamw_state_dict_round_trip(dev):
    """moments + step survive state_dict()/load_state_dict(): a resumed run continues bit for bit."""
    from slam_llm_amd.model import SlamAdamW, SlamHipModel
    cfg = dict(O.make_config(), lora_dropout=0.0)
    W = O.init_weights(cfg, seed=42)
    ob = O.synth_batch(cfg, O.synth_audio(2, 1.0, seed=5), prompt_len=4, answer_lens=(3, 6), seed=6, left_pad=True, pad_to_30s=False)
    gb = {k: v.to(dev) for k, v in ob.items()}

    def steps(model, opt, n):
        for _ in range(n):
            out, _ = model(**{k: v.clone() for k, v in gb.items()})
            out.loss.backward()
            opt.step()
            opt.zero_grad()
    a = SlamHipModel(dict(cfg), dev).load_weights(W).train()
    oa = SlamAdamW(a, lr=1e-2, weight_decay=0.01)
    steps(a, oa, 2)
    sd_model = {k: v.clone() for k, v in a.state_dict().items()}
    sd_opt = oa.state_dict()
    assert sd_opt["slam"]["step"] == 2 and float(sd_opt["slam"]["exp_avg_sq"].abs().max()) > 0
    steps(a, oa, 1)
    b = SlamHipModel(dict(cfg), dev).load_weights(W).train()
    b.load_state_dict(sd_model, strict=False)
    b.mark_params_updated()
    ob_ = SlamAdamW(b, lr=1e-2, weight_decay=0.01)
    ob_.load_state_dict(sd_opt)
    steps(b, ob_, 1)
    assert torch.equal(a.store.flat, b.store.flat)


@pytest.mark.timeout(1500)
def test_true_width_step_matches_oracle(dev):
    """the headline widths had no parity test.  Whisper-large-v3 widths (128 mel, d 1280, 20 heads) x 1
    layer -> Llama-3-8B widths (d 4096, 32q/8kv heads of 128, ffn 14336, rope theta 5e5) x 2 layers, V = 128 256, LoRA r16 on
    q,v, linear projector 6400 -> 2048 -> 4096; B = 2 short ragged clips, right-padded (aispeech collator).  K = 4160 / 14336 /
    28672 products, the 128 256-wide CE and the multi-chunk lm_head path (chunk rows forced below M) against the fp32
    oracle: loss abs <= 1e-2, accuracy exact up to one token, every trainable gradient cosine >= 0.999."""
    from slam_llm_amd.model import SlamHipModel, make_config
    cfg = make_config("whisper-large-v3", "llama-3-8b", enc_layers=1, llm_layers=2, lora_r=16, lora_alpha=32,
                      lora_targets=("q_proj", "v_proj"), lora_dropout=0.0)
    W = O.init_weights(cfg, seed=42)
    audio = O.synth_audio(2, 3.0, seed=1234)
    ob = O.synth_batch(cfg, audio, prompt_len=16, answer_lens=(24, 9), seed=1236, left_pad=False, pad_to_30s=False)
    names = O.trainable_names(W)
    for n in names:
        W[n].requires_grad_(True)
    torch.set_num_threads(min(64, os.cpu_count()))
    loss_ref, _, acc_ref, _ = O.slam_forward(W, cfg, ob)
    loss_ref.backward()
    grads = {n: W[n].grad.detach().clone() for n in names}
    for n in names:
        W[n].requires_grad_(False)
        W[n].grad = None
    model = SlamHipModel(dict(cfg), dev).load_weights(W)
    del W
    model.train()
    M = ob["input_ids"].numel()
    model.llm.lm_head_chunk_rows = max(16, (M // 3) // 16 * 16)          # three row chunks, like the C3 batch
    gb = {k: v.to(dev) for k, v in ob.items()}
    outputs, acc = model(**gb)
    outputs.loss.backward()
    n_valid = int((ob["labels"][:, 1:] != -100).sum())
    assert abs(float(outputs.loss) - float(loss_ref)) <= 1e-2, (float(outputs.loss), float(loss_ref))
    assert abs(float(acc) - float(acc_ref)) <= 1.0 / n_valid + 1e-6
    worst = 1.0
    for n, p in model.store.params.items():
        cs = G.cosine(grads[n].numpy(), p.grad.float().cpu().numpy())
        worst = min(worst, cs)
        G.floor_check(cs, 0.999, f"grad {n}: cosine {cs}")
        gn, mn = float(grads[n].norm()), float(p.grad.float().norm())
        assert abs(mn - gn) <= 3e-2 * gn + 1e-9, f"grad {n}: norm {mn} vs {gn}"
    print(f"true-width step: loss {float(outputs.loss):.4f} vs {float(loss_ref):.4f}, worst gradient cosine {worst:.6f}")


def test_peft_ckpt_adapter_directory_round_trip(dev, tmp_path):
    """train a few steps, write the adapter the way peft's save_pretrained does, reload it through model_factory(peft_ckpt=...)
    (slam_model.py:210-213): the adapter's own config wins over train_config.peft_config, LoRA tensors and the loss are identical"""
    from slam_llm_amd.slam_model_hip import model_factory, save_peft_adapter
    tc, mc, _ = _tiny_recipe(peft=dict(r=16, lora_alpha=32, target_modules=["q_proj", "k_proj", "v_proj", "o_proj"], lora_dropout=0.0))
    model, _ = model_factory(tc, mc)
    model.train()
    cfg = model.cfg
    ob = O.synth_batch(cfg, O.synth_audio(2, 1.0, seed=5), prompt_len=4, answer_lens=(3, 6), seed=6, left_pad=True, pad_to_30s=False)
    gb = {k: v.to(dev) for k, v in ob.items()}
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    for _ in range(2):
        out, _ = model(**{k: v.clone() for k, v in gb.items()})
        out.loss.backward()
        opt.step()
        opt.zero_grad()
    keys = save_peft_adapter(model, str(tmp_path / "adapter"))
    assert "base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight" in keys and len(keys) == 2 * 4 * cfg["llm_layers"]
    torch.save({k: v for k, v in model.state_dict().items() if k.startswith("encoder_projector.")}, tmp_path / "proj.pt")
    tc2, mc2, _ = _tiny_recipe(peft=dict(r=8))          # the recipe says r = 8 on q,v: the adapter directory must win
    m2, _ = model_factory(tc2, mc2, peft_ckpt=str(tmp_path / "adapter"), ckpt_path=str(tmp_path / "proj.pt"))
    assert m2.cfg["lora_r"] == 16 and m2.cfg["lora_targets"] == ("q_proj", "k_proj", "v_proj", "o_proj")
    for k, v in model.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    model.eval(); m2.eval()
    with torch.no_grad():
        a, _ = model(**{k: v.clone() for k, v in gb.items()})
        b, _ = m2(**{k: v.clone() for k, v in gb.items()})
    assert float(a.loss) == float(b.loss)
