"""The drop-in seam, on CPU: the plugin file loaded the way the reference loads it, fed the reference's own
recipe config defaults, and the two dataset formats.

* loader: `model_config.file = "<path>.py:model_factory"` / `dataset_config.file = "<path>.py:get_speech_dataset"` resolved by
  src/slam_llm/utils/model_utils.py:4-29 / utils/dataset_utils.py:14-57 (split on ":", require a .py file on disk,
  SourceFileLoader by path, getattr).  When /root/reference is present (authoring container) the reference's OWN loader
  source is executed (loaded by path: its package __init__ pulls deepspeed); everywhere else a restatement is used.
* configs: tests/golden/ref_configs.json = the dataclass defaults of examples/asr_librispeech/asr_config.py:8-130 and
  examples/aispeech_asr/aispeech_asr_config.py:7-143 (oracle/make_golden_configs.py), plus the `++a.b=v` overrides the five
  BASELINE.json configs imply.
"""
import importlib.machinery
import importlib.util
import json
import logging
import os
import struct
import sys
import types
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import slam_oracle as O
from tests import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGIN = os.path.join(ROOT, "slam_llm_amd", "slam_model_hip.py")
REF_UTILS = "/root/reference/src/slam_llm/utils"


class Cfg(dict):
    """attribute + .get access like the OmegaConf DictConfig nodes the reference passes around (finetune.py:75-88)"""
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return Cfg(v) if isinstance(v, dict) else v

    def get(self, k, d=None):
        v = dict.get(self, k, d)
        return Cfg(v) if isinstance(v, dict) else v


def _restated_loaders():
    def load_module_from_py_file(py_file):                     # dataset_utils.py:14-25
        module_name = Path(py_file).name
        loader = importlib.machinery.SourceFileLoader(module_name, py_file)
        spec = importlib.util.spec_from_loader(module_name, loader)
        module = importlib.util.module_from_spec(spec)
        loader.exec_module(module)
        return module

    def get_custom_model_factory(model_config, logger):        # model_utils.py:4-29
        path = model_config.get("file", None)
        module_path, func_name = path.split(":") if ":" in path else (path, "model_factory")
        if not module_path.endswith(".py"):
            raise ValueError(f"Dataset file {module_path} is not a .py file.")
        if not Path(module_path).is_file():
            raise FileNotFoundError(module_path)
        return getattr(load_module_from_py_file(Path(module_path).as_posix()), func_name)

    def get_custom_dataset(dataset_config, tokenizer, split):  # dataset_utils.py:28-46
        module_path, func_name = dataset_config.file.split(":") if ":" in dataset_config.file else (dataset_config.file, "get_custom_dataset")
        if not module_path.endswith(".py"):
            raise ValueError(f"Dataset file {module_path} is not a .py file.")
        if not Path(module_path).is_file():
            raise FileNotFoundError(module_path)
        return getattr(load_module_from_py_file(Path(module_path).as_posix()), func_name)(dataset_config, tokenizer, split)
    return get_custom_model_factory, get_custom_dataset, "restated"


def loaders():
    """(get_custom_model_factory, get_custom_dataset, origin)"""
    if not os.path.isdir(REF_UTILS):
        return _restated_loaders()
    spec = importlib.util.spec_from_file_location("ref_dataset_utils", os.path.join(REF_UTILS, "dataset_utils.py"))
    du = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(du)
    saved = {k: sys.modules.get(k) for k in ("slam_llm", "slam_llm.utils", "slam_llm.utils.dataset_utils")}
    try:   # model_utils.py:1 does `from slam_llm.utils.dataset_utils import load_module_from_py_file`
        sys.modules["slam_llm"] = types.ModuleType("slam_llm")
        sys.modules["slam_llm.utils"] = types.ModuleType("slam_llm.utils")
        sys.modules["slam_llm.utils.dataset_utils"] = du
        spec = importlib.util.spec_from_file_location("ref_model_utils", os.path.join(REF_UTILS, "model_utils.py"))
        mu = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mu)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mu.get_custom_model_factory, du.get_custom_dataset, "reference"


def ref_defaults(recipe):
    with open(os.path.join(G.GOLD, "ref_configs.json")) as f:
        return json.load(f)[recipe]


def recipe_configs(recipe, model=None, train=None, peft=None, data=None):
    d = ref_defaults(recipe)
    tc = dict(d["TrainConfig"])
    tc["peft_config"] = dict(tc["peft_config"], **(peft or {}))
    tc.update(train or {})
    return Cfg(tc), Cfg(dict(d["ModelConfig"], **(model or {}))), Cfg(dict(d["DataConfig"], **(data or {})))


def test_plugin_entry_points_resolve_through_the_reference_loader():
    get_factory, get_dataset, origin = loaders()
    logger = logging.getLogger("test")
    factory = get_factory(Cfg(file=PLUGIN + ":model_factory"), logger)
    assert callable(factory) and factory.__name__ == "model_factory"
    assert get_factory(Cfg(file=PLUGIN), logger).__name__ == "model_factory"       # default function name (model_utils.py:15-16)
    with pytest.raises(ValueError):
        get_factory(Cfg(file=PLUGIN[:-3] + ":model_factory"), logger)
    with pytest.raises(FileNotFoundError):
        get_factory(Cfg(file="/nonexistent/x.py:model_factory"), logger)
    with pytest.raises(AttributeError):
        get_factory(Cfg(file=PLUGIN + ":no_such_factory"), logger)
    print("loader origin:", origin)


# the `++` overrides each BASELINE.json config implies on top of the recipe defaults
BASELINE_CONFIGS = {
    "C1": ("asr_librispeech", dict(encoder_name="whisper", encoder_path="/ckpt/Whisper/tiny.pt", encoder_dim=384, llm_name="TinyLlama-1.1B", llm_dim=2048),
           dict(use_peft=True, freeze_encoder=True, freeze_llm=True), dict(r=8),
           dict(enc_dim=384, enc_layers=4, n_mels=80, llm_dim=2048, llm_layers=22, llm_kv_heads=4, vocab=32000, lora_r=8, lora_targets=("q_proj", "v_proj"))),
    "C2": ("asr_librispeech", dict(encoder_name="whisper", encoder_path="/ckpt/Whisper/base.pt", encoder_dim=512, llm_name="Meta-Llama-3-8B"),
           dict(use_peft=True, freeze_encoder=True, freeze_llm=True), dict(r=16),
           dict(enc_dim=512, enc_layers=6, llm_dim=4096, llm_layers=32, llm_kv_heads=8, llm_head_dim=128, llm_ffn=14336, vocab=128256, rope_theta=500000.0, lora_r=16)),
    "C3": ("aispeech_asr", dict(encoder_name="whisper", encoder_path="/ckpt/Whisper/large-v3.pt", llm_name="llama-3-8b"),
           dict(use_peft=True, freeze_encoder=True, freeze_llm=True, enable_ddp=True), dict(r=16, lora_alpha=32, target_modules=["q_proj", "v_proj"]),
           dict(enc_dim=1280, enc_layers=32, enc_heads=20, n_mels=128, llm_dim=4096, vocab=128256, lora_r=16, projector="linear", ds_rate=5)),
    "C4": ("asr_librispeech", dict(encoder_name="hubert", encoder_path="/ckpt/hubert_large_ll60k.pt", encoder_dim=1024, encoder_projector="q-former",
                                    query_len=32, qformer_layers=8, llm_name="vicuna-7b-v1.5"),
           dict(use_peft=True, freeze_encoder=True, freeze_llm=True), dict(r=32),
           dict(encoder_name="hubert", hub_dim=1024, hub_layers=24, hub_heads=16, enc_dim=1024, projector="q-former", qf_queries=32, qf_layers=8,
                llm_kv_heads=32, llm_ffn=11008, vocab=32000, lora_r=32)),
    "C5": ("aispeech_asr", dict(encoder_name="whisper", encoder_path="/ckpt/Whisper/large-v3.pt", llm_name="llama-3-8b"),
           dict(use_peft=True, freeze_encoder=True, freeze_llm=True, enable_ddp=True), dict(),
           # recipe default LoRA of aispeech_asr_config.py:32-40: r 64, alpha 16, all seven projections
           dict(enc_dim=1280, lora_r=64, lora_alpha=16.0,
                lora_targets=("q_proj", "k_proj", "v_proj", "o_proj", "up_proj", "gate_proj", "down_proj"), lora_dropout=0.05)),
}


@pytest.mark.parametrize("name", sorted(BASELINE_CONFIGS))
def test_build_config_on_reference_recipe_defaults(name):
    from slam_llm_amd.slam_model_hip import build_config, check_supported
    recipe, model_over, train_over, peft_over, expect = BASELINE_CONFIGS[name]
    tc, mc, _ = recipe_configs(recipe, model_over, train_over, peft_over)
    check_supported(tc, mc)
    cfg = build_config(tc, mc)
    for k, v in expect.items():
        assert cfg[k] == v, (name, k, cfg[k], v)
    if name == "C3":   # 21.5 M projector + 6.8 M LoRA = 28.3 M trainable parameters
        d, k5, hid, dl, r, L = cfg["enc_dim"], cfg["ds_rate"], cfg["proj_hidden"], cfg["llm_dim"], cfg["lora_r"], cfg["llm_layers"]
        proj = k5 * d * hid + hid + hid * dl + dl
        lora = L * r * ((dl + dl) + (dl + cfg["llm_kv_heads"] * cfg["llm_head_dim"]))
        assert proj == 21_501_952 and lora == 6_815_744


def test_untouched_recipe_defaults_are_rejected_loudly():
    """the dataclass defaults alone describe no speech model (encoder_name None, vicuna-13b with llm_dim 4096, freeze_encoder
    False): the plugin must say so instead of guessing"""
    from slam_llm_amd.slam_model_hip import build_config, check_supported
    tc, mc, _ = recipe_configs("asr_librispeech")
    with pytest.raises(NotImplementedError, match="freeze_encoder"):
        check_supported(tc, mc)
    with pytest.raises(NotImplementedError, match="encoder_name"):
        build_config(tc, mc)
    tc, mc, _ = recipe_configs("asr_librispeech", dict(encoder_name="whisper", encoder_path="large-v2.pt"), dict(use_peft=True))
    with pytest.raises(ValueError, match="llm_dim"):      # vicuna-13b-v1.5 is 5120 wide, the default llm_dim says 4096
        build_config(tc, mc)
    tc, mc, _ = recipe_configs("aispeech_asr", dict(encoder_name="beats"), dict(freeze_encoder=True))
    with pytest.raises(NotImplementedError, match="beats"):
        build_config(tc, mc)
    # unfrozen encoders: Whisper (any projector), HuBERT and WavLM are served (the base geometries refuse at model construction)
    for enc, proj in (("whisper", "q-former"), ("whisper", "cov1d-linear"), ("hubert", "linear"), ("wavlm", "linear")):
        tc, mc, _ = recipe_configs("aispeech_asr", dict(encoder_name=enc, encoder_projector=proj), dict(freeze_encoder=False, use_peft=True))
        check_supported(tc, mc)
    tc, mc, _ = recipe_configs("aispeech_asr", dict(encoder_name="whisper"), dict(freeze_encoder=True, use_peft=True, enable_deepspeed=True))
    with pytest.raises(NotImplementedError, match="DeepSpeed"):
        check_supported(tc, mc)


def test_wavlm_and_unfrozen_whisper_recipes_build():
    """aispeech_asr with the WavLM-Large encoder (models/slam_model.py:89-91, :333-334) and a Whisper recipe with
    train_config.freeze_encoder=false map to supported configurations"""
    from slam_llm_amd.slam_model_hip import build_config, check_supported
    tc, mc, _ = recipe_configs("aispeech_asr", dict(encoder_name="wavlm", encoder_path="/ckpt/WavLM-Large.pt", encoder_dim=1024, llm_name="vicuna-7b-v1.5",
                                                     llm_dim=4096, encoder_projector="linear", normalize=True),
                               dict(freeze_encoder=True, use_peft=True))
    check_supported(tc, mc)
    cfg = build_config(tc, mc)
    assert cfg["encoder_name"] == "wavlm" and cfg["enc_dim"] == 1024 and cfg["hub_layers"] == 24 and cfg["wavlm_buckets"] == 320 \
        and cfg["wavlm_max_distance"] == 800 and cfg["freeze_encoder"] is True
    # the train-mode regularisers of the un-frozen wave encoder arrive with the reference module's defaults (WavLMConfig, WavLM.py:180-185)
    # and are overridable per knob
    assert (cfg["hub_dropout"], cfg["hub_attention_dropout"], cfg["hub_activation_dropout"], cfg["hub_dropout_input"], cfg["hub_layerdrop"]) == \
        (0.1, 0.1, 0.0, 0.0, 0.0)
    tc2, mc2, _ = recipe_configs("aispeech_asr", dict(encoder_name="wavlm", encoder_path="/ckpt/WavLM-Large.pt", encoder_dim=1024, llm_name="vicuna-7b-v1.5",
                                                       llm_dim=4096, encoder_projector="linear", normalize=True, encoder_dropout=0.0,
                                                       encoder_layerdrop=0.05), dict(freeze_encoder=False, use_peft=True))
    cfg2 = build_config(tc2, mc2)
    assert cfg2["hub_dropout"] == 0.0 and cfg2["hub_layerdrop"] == 0.05 and cfg2["hub_attention_dropout"] == 0.1
    tc, mc, _ = recipe_configs("asr_librispeech", dict(encoder_name="whisper", encoder_path="/ckpt/large-v3.pt", encoder_dim=1280, llm_name="llama-3-8b",
                                                        llm_dim=4096, encoder_projector="linear"), dict(freeze_encoder=False, use_peft=True))
    check_supported(tc, mc)
    assert build_config(tc, mc)["freeze_encoder"] is False


def test_preset_guess_uses_the_basename_first_and_the_recipes_spellings():
    """the reference HuBERT recipe's checkpoint is `hubert_xtralarge_ll60k_finetune_ls960.pt` (xtralarge = xlarge), and a
    directory component must not outvote the file name; find_unused_parameters=true is refused loudly (DDP cannot see through the
    one autograd node that produces every gradient)."""
    from slam_llm_amd.slam_model_hip import HUBERT_PRESETS, WAVLM_PRESETS, _guess_preset, check_supported
    assert _guess_preset("/nfs/ckpt/hubert_xtralarge_ll60k_finetune_ls960.pt", HUBERT_PRESETS) == "hubert-xlarge"
    assert _guess_preset("/hubert-large-models/hubert_base_ls960.pt", HUBERT_PRESETS) == "hubert-base"
    assert _guess_preset("/ckpt/hubert_large_ll60k.pt", HUBERT_PRESETS) == "hubert-large"
    assert _guess_preset("/wavlm-large/WavLM-Base+.pt", WAVLM_PRESETS) == "wavlm-base"
    # a generic file name falls back to its parent directories, nearest first: fairseq work dirs, HF snapshot dirs
    assert _guess_preset("/hubert-large/model.pt", HUBERT_PRESETS) == "hubert-large"
    assert _guess_preset("/ckpt/hubert_large_ll60k/checkpoint_best.pt", HUBERT_PRESETS) == "hubert-large"
    assert _guess_preset("/hub/models--facebook--hubert-base-ls960/snapshots/abc/model.safetensors", HUBERT_PRESETS) == "hubert-base"
    with pytest.raises(ValueError, match="arch_encoder"):
        _guess_preset("/data/run17/checkpoint_best.pt", HUBERT_PRESETS)
    with pytest.raises(NotImplementedError, match="find_unused_parameters"):
        check_supported(dict(use_peft=True, freeze_encoder=True, find_unused_parameters=True), dict(encoder_name="whisper"))


def test_base_encoder_geometries_are_selected_from_the_checkpoint_name():
    """WavLM Base / Base+ and HuBERT-base (group-norm extractor, post-LN layers, 12 x 768) through `encoder_path`; the large
    presets keep the layer-norm extractor / pre-LN defaults"""
    from slam_llm_amd.slam_model_hip import build_config, check_supported
    for path, name, dim, layers, post_ln in (("/ckpt/WavLM-Base+.pt", "wavlm", 768, 12, True), ("/ckpt/WavLM-Base.pt", "wavlm", 768, 12, True),
                                              ("/ckpt/wavlm_base_plus.pt", "wavlm", 768, 12, True), ("/ckpt/WavLM-Large.pt", "wavlm", 1024, 24, False),
                                              ("/ckpt/hubert_base_ls960.pt", "hubert", 768, 12, True), ("/ckpt/hubert_large_ll60k.pt", "hubert", 1024, 24, False)):
        tc, mc, _ = recipe_configs("aispeech_asr", dict(encoder_name=name, encoder_path=path, encoder_dim=dim, llm_name="vicuna-7b-v1.5",
                                                         llm_dim=4096, encoder_projector="linear"), dict(freeze_encoder=True, use_peft=True))
        check_supported(tc, mc)
        cfg = build_config(tc, mc)
        assert cfg["enc_dim"] == dim and cfg["hub_dim"] == dim and cfg["hub_layers"] == layers and cfg["hub_dim"] // cfg["hub_heads"] == 64, path
        assert (cfg.get("hub_extractor_mode", "layer_norm") == "default") == post_ln, path
        assert cfg.get("hub_layer_norm_first", True) == (not post_ln), path


# ---------------------------------------------------------------------------------------------- dataset formats
def _wav(path, pcm, rate=16000):
    data = pcm.astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16)
                + b"data" + struct.pack("<I", len(data)) + data)


class MergingTok:
    """tokenizer stub whose encoding of prompt+answer is NOT the concatenation of the two encodings: the character pair
    ':t' (end of 'ASSISTANT:' + first answer letter) merges into one id -- what a BPE tokenizer does at that boundary.  The
    reference encodes the whole string (speech_dataset.py:137-139)."""
    eos_token_id, pad_token_id = 2, 0

    def encode(self, text):
        ids, i = [1], 0
        while i < len(text):
            if text[i:i + 2] == ":t":
                ids.append(99)
                i += 2
            else:
                ids.append(3 + ord(text[i]) % 50)
                i += 1
        return ids

    def batch_decode(self, tokens, **kw):
        return ["".join(chr(97 + int(t) % 26) for t in row) for row in tokens]


def _jsonl_dataset(tmp_path, get_dataset, input_type, normalize=False, inference=False):
    g = torch.Generator().manual_seed(3)
    clips = [(torch.randn(n, generator=g) * 2000).round().clamp(-32768, 32767).numpy().astype(np.int16) for n in (16000, 9600, 24000)]
    rows = []
    for i, pcm in enumerate(clips):
        _wav(tmp_path / f"u{i}.wav", pcm)
        rows.append({"key": f"u{i}", "source": str(tmp_path / f"u{i}.wav"), "target": f"t{i} words"})
    (tmp_path / "train.jsonl").write_text("\n".join(json.dumps(r) for r in rows) + "\n")
    _, _, dc = recipe_configs("asr_librispeech", data=dict(file=PLUGIN + ":get_speech_dataset", train_data_path=str(tmp_path / "train.jsonl"),
                                                           val_data_path=str(tmp_path / "train.jsonl"), input_type=input_type,
                                                           normalize=normalize, inference_mode=inference))
    return get_dataset(dc, MergingTok(), "train"), clips, rows


@pytest.mark.parametrize("input_type", ["mel", "raw"])
def test_jsonl_dataset_matches_reference_layout(tmp_path, input_type):
    """asr_librispeech JSONL format through the reference's dataset loader: token layout of speech_dataset.py:86-161 (whole-string
    encoding, label masking), left/right padding collator :216-291, per-input_type audio fields and placeholder counts."""
    _, get_dataset, _ = loaders()
    ds, clips, rows = _jsonl_dataset(tmp_path, get_dataset, input_type, normalize=(input_type == "raw"))
    tok = MergingTok()
    prompt = "USER: {}\n ASSISTANT:".format("Transcribe speech to text. Output the transcription directly without redundant content. "
                                            "Ensure that the output is not duplicated. ")
    samples = [ds[i] for i in range(len(ds))]
    ref_samples = []
    for s, pcm, row in zip(samples, clips, rows):
        alen = (len(pcm) // 320 // 5) if input_type == "raw" else 300          # speech_dataset.py:98-99 / :104-105 (30 s pad -> 3000 frames)
        pids = tok.encode(prompt)
        whole = tok.encode(prompt + row["target"])
        assert whole[: len(pids)] != pids or 99 in whole                       # the boundary merge is live in this stub
        ids = torch.tensor([-1] * alen + whole + [2])
        labels = ids.clone()
        labels[: alen + len(pids)] = -100
        assert torch.equal(s["input_ids"], ids) and torch.equal(s["labels"], labels) and s["audio_length"] == alen
        wav = torch.from_numpy(pcm.astype(np.float32) / 32768)
        if input_type == "raw":
            wav = torch.nn.functional.layer_norm(wav, wav.shape)
        assert torch.allclose(s["audio"], wav)
        ref_samples.append({"input_ids": ids.clamp(min=0), "labels": labels, "attention_mask": torch.ones_like(ids).bool(),
                            "audio_length": alen, "prompt_length": len(pids)})
    batch = ds.collator(samples)
    refb = O.collate_left_pad(ref_samples, pad_id=tok.pad_token_id)
    for k in ("labels", "attention_mask", "modality_mask"):
        assert torch.equal(batch[k].long(), refb[k].long()), k
    assert torch.equal(batch["input_ids"].clamp(min=0).masked_fill(batch["modality_mask"], 0), refb["input_ids"].masked_fill(refb["modality_mask"].bool(), 0))
    if input_type == "raw":
        assert batch["audio"].shape == (3, 24000) and batch["audio_mask"].sum(1).tolist() == [16000, 9600, 24000]
        assert batch["audio_mask"].dtype == torch.float32
    else:
        assert "audio_mask" not in batch and batch["audio_len"].tolist() == [16000, 9600, 24000]


def test_jsonl_inference_mode_and_dataloader_contract(tmp_path):
    """inference_mode samples carry key/target ([audio, prompt] only, speech_dataset.py:120-134); the DataLoader the reference
    builds for the `custom` strategy (utils/config_utils.py:100-112: collate_fn=dataset.collator) works on the object"""
    _, get_dataset, _ = loaders()
    ds, clips, rows = _jsonl_dataset(tmp_path, get_dataset, "mel", inference=True)
    dl = torch.utils.data.DataLoader(ds, batch_size=2, collate_fn=ds.collator)
    batches = list(dl)
    assert [b["keys"] for b in batches] == [["u0", "u1"], ["u2"]] and batches[0]["targets"][1] == "t1 words"
    prompt = "USER: {}\n ASSISTANT:".format("Transcribe speech to text. Output the transcription directly without redundant content. "
                                            "Ensure that the output is not duplicated. ")
    assert "labels" not in batches[0] and batches[0]["input_ids"].shape[1] == 300 + len(MergingTok().encode(prompt))


def test_dataset_rejects_missing_input_type(tmp_path):
    _, get_dataset, _ = loaders()
    with pytest.raises(ValueError, match="input_type"):
        _jsonl_dataset(tmp_path, get_dataset, None)


def test_peft_adapter_directory_key_mapping(tmp_path):
    """`peft_ckpt` (slam_model.py:210-213): adapter_config.json defines the LoRA, adapter_model.bin keys have the adapter name
    stripped (peft 0.6.0 get_peft_model_state_dict) -> mapped to this model's state_dict keys"""
    from slam_llm_amd.slam_model_hip import read_peft_adapter
    d = tmp_path / "adapter"
    d.mkdir()
    (d / "adapter_config.json").write_text(json.dumps(dict(peft_type="LORA", r=4, lora_alpha=8, lora_dropout=0.1, bias="none",
                                                           target_modules=["q_proj", "v_proj"], task_type="CAUSAL_LM")))
    sd = {f"base_model.model.model.layers.{i}.self_attn.{m}.lora_{ab}.weight": torch.full((2, 2), float(i))
          for i in range(2) for m in ("q_proj", "v_proj") for ab in "AB"}
    torch.save(sd, d / "adapter_model.bin")
    lora, state = read_peft_adapter(str(d))
    assert lora == dict(lora_r=4, lora_alpha=8.0, lora_targets=("q_proj", "v_proj"), lora_dropout=0.1)
    assert sorted(state)[0] == "llm.base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight" and len(state) == 8
    with pytest.raises(FileNotFoundError):
        read_peft_adapter(str(tmp_path))
    (d / "adapter_config.json").write_text(json.dumps(dict(peft_type="PREFIX_TUNING", r=4, lora_alpha=8, target_modules=[])))
    with pytest.raises(NotImplementedError):
        read_peft_adapter(str(d))
