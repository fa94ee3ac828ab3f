"""Plugin file for the reference's loaders: use with

    ++model_config.file=<repo>/slam_llm_amd/slam_model_hip.py:model_factory
    ++dataset_config.file=<repo>/slam_llm_amd/slam_model_hip.py:get_speech_dataset

`model_factory(train_config, model_config, **kwargs) -> (model, tokenizer)` has the signature of
src/slam_llm/models/slam_model.py:21-51 and is resolved by src/slam_llm/utils/model_utils.py:4-29 (path.py:func).
Configuration is read with `.get()` so both the asr_librispeech and the aispeech_asr config dataclasses work
(SURVEY g14).  Weights: `model_config.encoder_path` / `llm_path` may point at *.pt / *.safetensors state dicts in
the reference's key names; with `model_config.random_init=true` (benchmarks, tests) seeded random weights at the
true dimensions are generated directly in HBM.
"""
from __future__ import annotations

import logging
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
if os.path.dirname(_HERE) not in sys.path:
    sys.path.insert(0, os.path.dirname(_HERE))

from slam_llm_amd.model import PRESETS, SlamHipModel, make_config  # noqa: E402

logger = logging.getLogger(__name__)


def _get(cfg, key, default=None):
    if cfg is None:
        return default
    if hasattr(cfg, "get"):
        v = cfg.get(key, default)
        return default if v is None else v
    return getattr(cfg, key, default)


PRESET_ALIASES = {"xtralarge": "xlarge", "extralarge": "xlarge", "x-large": "xlarge", "xtra-large": "xlarge"}


def _guess_preset(name: str, table) -> str:
    """architecture preset from a checkpoint name.  The BASENAME votes first (a directory such as `/hubert-large-models/` must not
    pick the geometry of `hubert_base_ls960.pt` inside it); the reference recipes' spellings are normalised first
    (`hubert_xtralarge_ll60k_finetune_ls960.pt`, examples/asr_librispeech/scripts/finetune_hubert_xtralarge_linear_vicuna_7b.sh)."""
    parts = [c for c in (name or "").replace("\\", "/").rstrip("/").split("/") if c]
    # the basename votes first; a generic file name (`checkpoint_best.pt`, `model.safetensors` of an HF snapshot) falls back to its parent
    # directories, nearest first (`/ckpt/hubert_large_ll60k/checkpoint_best.pt`)
    for comp in reversed(parts or [""]):
        n = comp.lower().replace("_", "-")
        for a, b in PRESET_ALIASES.items():
            n = n.replace(a, b)
        for k in sorted(table, key=len, reverse=True):
            if k in n or k.replace("-", "") in n.replace("-", ""):
                return k
    raise ValueError(f"cannot map '{name}' to a known architecture preset {sorted(table)}; pass model_config.arch_encoder / arch_llm explicitly")


HUBERT_PRESETS = {
    # fairseq / HF HubertConfig geometry
    "hubert-large": dict(hub_conv_dim=(512,) * 7, hub_conv_kernel=(10, 3, 3, 3, 3, 2, 2), hub_conv_stride=(5, 2, 2, 2, 2, 2, 2),
                         hub_dim=1024, hub_heads=16, hub_layers=24, hub_ffn=4096, hub_pos_k=128, hub_pos_groups=16, hub_eps=1e-5),
    "hubert-xlarge": dict(hub_conv_dim=(512,) * 7, hub_conv_kernel=(10, 3, 3, 3, 3, 2, 2), hub_conv_stride=(5, 2, 2, 2, 2, 2, 2),
                          hub_dim=1280, hub_heads=20, hub_layers=48, hub_ffn=5120, hub_pos_k=128, hub_pos_groups=16, hub_eps=1e-5),
    # base: "default" extractor (GroupNorm over time after the first conv only, no conv bias), post-LN layers
    "hubert-base": dict(hub_conv_dim=(512,) * 7, hub_conv_kernel=(10, 3, 3, 3, 3, 2, 2), hub_conv_stride=(5, 2, 2, 2, 2, 2, 2),
                        hub_dim=768, hub_heads=12, hub_layers=12, hub_ffn=3072, hub_pos_k=128, hub_pos_groups=16, hub_eps=1e-5,
                        hub_extractor_mode="default", hub_layer_norm_first=False),
}

WAVLM_PRESETS = {
    # WavLM-Large (the released checkpoint's cfg: extractor_mode layer_norm, conv_bias false, layer_norm_first, gru_rel_pos,
    # 320 buckets / max_distance 800; models/wavlm/WavLM.py:162-214).
    "wavlm-large": dict(HUBERT_PRESETS["hubert-large"], wavlm_buckets=320, wavlm_max_distance=800),
    # Base / Base+ (same architecture, different training data): extractor_mode "default" (GroupNorm after the first conv only),
    # post-LN layers, 12 x 768 / 12 heads / ffn 3072
    "wavlm-base": dict(HUBERT_PRESETS["hubert-large"], hub_dim=768, hub_heads=12, hub_layers=12, hub_ffn=3072, wavlm_buckets=320,
                       wavlm_max_distance=800, hub_extractor_mode="default", hub_layer_norm_first=False),
}
WAVLM_PRESETS["wavlm-base-plus"] = WAVLM_PRESETS["wavlm-base"]


def build_config(train_config, model_config) -> dict:
    enc_name = _get(model_config, "encoder_name", None)
    if enc_name not in ("whisper", "hubert", "wavlm"):
        # the recipe dataclasses default encoder_name to None (asr_config.py:14: a text-only LLM in the reference,
        # slam_model.py:68-116 returns no encoder); this plugin is the speech path only
        raise NotImplementedError(f"model_config.encoder_name={enc_name!r}: the HIP path covers the Whisper (slam_model.py:320-321), "
                                  "HuBERT (:335-341) and WavLM (:333-334) branches")
    projector = _get(model_config, "encoder_projector", "linear")
    if projector not in ("linear", "cov1d-linear", "q-former"):
        raise NotImplementedError("encoder_projector must be `linear` (EncoderProjectorConcat), `cov1d-linear` "
                                  "(EncoderProjectorCov1d) or `q-former` (EncoderProjectorQFormer)")
    enc_presets = {k: v for k, v in PRESETS.items() if k.startswith("whisper")}
    llm_presets = {k: v for k, v in PRESETS.items() if not k.startswith("whisper")}
    llm = _get(model_config, "arch_llm") or _guess_preset(str(_get(model_config, "llm_name", "")), llm_presets)
    peft = _get(train_config, "peft_config", None)
    use_peft = bool(_get(train_config, "use_peft", False))
    # ++model_config.pad_or_trim=false: per-clip mel (aispeech recipes); ++model_config.varlen=true: right-padded batches run
    # the LLM on packed sequences (no pad tokens) -- identical results on every valid token, see DESIGN.md
    extra = dict(encoder_name=enc_name, projector=projector, pad_or_trim=bool(_get(model_config, "pad_or_trim", True)),
                 varlen=bool(_get(model_config, "varlen", False)),
                 # ++model_config.varlen_encoder=true (with pad_or_trim=false): ragged clips are encoded without pad frames, each
                 # exactly as if alone in the batch -- a stated deviation from the reference's zero-padded batch
                 varlen_encoder=bool(_get(model_config, "varlen_encoder", False)),
                 # HuBERT + Q-Former on ragged batches: the reference forwards fairseq's padding mask un-inverted;
                 # default = reference behaviour, true = attend to the real frames
                 hubert_qformer_mask_fix=bool(_get(model_config, "hubert_qformer_mask_fix", False)),
                 # train_config.freeze_encoder=false: the (Whisper) encoder trains with the projector and the adapters
                 freeze_encoder=bool(_get(train_config, "freeze_encoder", True)))
    if enc_name == "hubert":
        hp = _get(model_config, "arch_encoder") or _guess_preset(str(_get(model_config, "encoder_path", "hubert-large")).replace("_", "-"), HUBERT_PRESETS)
        extra.update(HUBERT_PRESETS[hp])
        extra["enc_dim"] = extra["hub_dim"]
        enc = None
    elif enc_name == "wavlm":
        hp = _get(model_config, "arch_encoder") or _guess_preset(str(_get(model_config, "encoder_path", "wavlm-large")).lower().replace("_", "-"), WAVLM_PRESETS)
        extra.update(WAVLM_PRESETS[hp])
        extra["enc_dim"] = extra["hub_dim"]
        enc = None
    else:
        enc = _get(model_config, "arch_encoder") or _guess_preset("whisper-" + str(_get(model_config, "encoder_path", "")).split("/")[-1].replace(".pt", ""), enc_presets)
    if enc_name in ("hubert", "wavlm"):
        # train-mode regularisers of the UN-FROZEN wave encoder (no effect when frozen or in eval mode): the reference module's own
        # defaults -- WavLMConfig (models/wavlm/WavLM.py:180-185; a raw WavLM checkpoint's `cfg` overrides them in model_factory) and
        # fairseq's HubertConfig carry the same numbers
        extra.update(hub_dropout=float(_get(model_config, "encoder_dropout", 0.1)),
                     hub_attention_dropout=float(_get(model_config, "encoder_attention_dropout", 0.1)),
                     hub_activation_dropout=float(_get(model_config, "encoder_activation_dropout", 0.0)),
                     hub_dropout_input=float(_get(model_config, "encoder_dropout_input", 0.0)),
                     hub_layerdrop=float(_get(model_config, "encoder_layerdrop", 0.0)))
    if projector == "q-former":
        extra.update(qf_dim=768, qf_heads=12, qf_ffn=3072, qf_eps=1e-12, qf_cross_freq=2,
                     qf_layers=int(_get(model_config, "qformer_layers", 8)), qf_queries=int(_get(model_config, "query_len", 64)),
                     qf_dropout=float(_get(model_config, "qformer_dropout", 0.1)))   # Blip2QFormerConfig() default, train mode only
    cfg = make_config(enc, llm,
                      ds_rate=int(_get(model_config, "encoder_projector_ds_rate", 5)),
                      lora_r=int(_get(peft, "r", 8)), lora_alpha=float(_get(peft, "lora_alpha", 32)),
                      lora_targets=tuple(_get(peft, "target_modules", ("q_proj", "v_proj"))) if use_peft else (),
                      lora_dropout=float(_get(peft, "lora_dropout", 0.05)) if use_peft else 0.0, **extra)
    overrides = _get(model_config, "arch_overrides", None)   # ++model_config.arch_overrides={llm_layers: 2, ...}: non-preset geometries
    if overrides:
        cfg.update({k: (tuple(v) if isinstance(v, list) else v) for k, v in dict(overrides).items()})
        if enc_name in ("hubert", "wavlm"):
            cfg["enc_dim"] = cfg["hub_dim"]
    if int(_get(model_config, "encoder_dim", cfg["enc_dim"])) != cfg["enc_dim"] or int(_get(model_config, "llm_dim", cfg["llm_dim"])) != cfg["llm_dim"]:
        raise ValueError(f"model_config.encoder_dim / llm_dim ({_get(model_config, 'encoder_dim')}, {_get(model_config, 'llm_dim')}) do not match "
                         f"the selected architecture ({cfg['enc_dim']}, {cfg['llm_dim']})")
    return cfg


def check_supported(train_config, model_config):
    """what the reference's factory would do with these flags that the HIP path does not implement -> loud errors
    (src/slam_llm/models/slam_model.py:68-221)."""
    if _get(train_config, "freeze_encoder", True) is False and _get(model_config, "encoder_name", None) not in ("whisper", "hubert", "wavlm"):
        raise NotImplementedError("train_config.freeze_encoder=false (unfrozen-encoder training, slam_model.py:110-113) is implemented for "
                                  "encoder_name=whisper, hubert (large / xlarge graph) and wavlm (WavLM-Large graph) -- hand-written encoder "
                                  "backward; linear, cov1d-linear and q-former projectors; otherwise pass ++train_config.freeze_encoder=true "
                                  "as the speech recipes do")
    if not bool(_get(train_config, "use_peft", False)) and _get(train_config, "freeze_llm", True) is False:
        raise NotImplementedError("full LLM fine-tuning (use_peft=false, freeze_llm=false) is out of scope: the HIP LLM is frozen + LoRA")
    if bool(_get(train_config, "quantization", False)) or bool(_get(train_config, "use_fast_kernels", False)):
        raise NotImplementedError("quantization / use_fast_kernels are library toggles of the reference stack (slam_model.py:145-146,187-197); "
                                  "the HIP path is bf16 with its own kernels")
    if bool(_get(train_config, "enable_fsdp", False)) or bool(_get(train_config, "enable_deepspeed", False)):
        raise NotImplementedError("FSDP / DeepSpeed are memory strategies the 288 GB part does not need (SURVEY 2e): run the "
                                  "HIP path with enable_ddp=true (DistributedDataParallel) or single-process")
    if bool(_get(train_config, "find_unused_parameters", False)):
        # finetune.py:183-184 forwards this flag to DDP.  Every trainable tensor receives a gradient in every backward of the HIP
        # step (one autograd node produces them all), so nothing is ever unused -- and DDP's unused-parameter search cannot see
        # through that node: it would mark all parameters unused up front and then fail with "marked ready twice".
        raise NotImplementedError("train_config.find_unused_parameters=true is neither needed nor supported on the HIP path (all "
                                  "trainable parameters get gradients every step); leave it false (the reference default)")
    peft = _get(train_config, "peft_config", None)
    if bool(_get(train_config, "use_peft", False)) and str(_get(peft, "peft_method", "lora")) != "lora":
        raise NotImplementedError("only peft_method=lora is implemented")


# ---------------------------------------------------------------------------------------------- peft adapter directories
PEFT_PREFIX = "llm."   # the LLM sits at `slam_model.llm` (models/slam_model.py:258), so peft's keys gain this prefix in model.state_dict()


def read_peft_adapter(adapter_dir: str):
    """`peft_ckpt` (src/slam_llm/models/slam_model.py:210-213: `PeftModel.from_pretrained(model, model_id=peft_ckpt,
    is_trainable=True)`): a peft 0.6.0 adapter directory = `adapter_config.json` (the LoraConfig that DEFINES the adapter: r,
    lora_alpha, target_modules, lora_dropout -- it wins over train_config.peft_config) + `adapter_model.safetensors` or
    `adapter_model.bin` whose keys carry no adapter name (`...q_proj.lora_A.weight`; peft's get_peft_model_state_dict strips
    `.default`).  Returns (lora settings for make_config, {state_dict key of this model: tensor})."""
    import json
    cfg_path = os.path.join(adapter_dir, "adapter_config.json")
    if not os.path.isfile(cfg_path):
        raise FileNotFoundError(f"peft_ckpt: {cfg_path} not found (expected a peft adapter directory)")
    with open(cfg_path) as f:
        ac = json.load(f)
    if str(ac.get("peft_type", "LORA")).upper() != "LORA":
        raise NotImplementedError(f"peft_ckpt: peft_type {ac.get('peft_type')!r} (only LORA adapters are implemented)")
    if ac.get("bias", "none") != "none" or ac.get("modules_to_save"):
        raise NotImplementedError("peft_ckpt: adapters with trainable biases / modules_to_save are not implemented")
    st_path, bin_path = os.path.join(adapter_dir, "adapter_model.safetensors"), os.path.join(adapter_dir, "adapter_model.bin")
    if os.path.isfile(st_path):
        from safetensors.torch import load_file
        raw = load_file(st_path)
    elif os.path.isfile(bin_path):
        raw = torch.load(bin_path, map_location="cpu")
    else:
        raise FileNotFoundError(f"peft_ckpt: neither adapter_model.safetensors nor adapter_model.bin in {adapter_dir}")
    state = {}
    for k, v in raw.items():
        for ab in ("lora_A", "lora_B"):
            if k.endswith(f".{ab}.weight"):
                k = k[: -len("weight")] + "default.weight"
        state[PEFT_PREFIX + k] = v
    lora = dict(lora_r=int(ac["r"]), lora_alpha=float(ac["lora_alpha"]), lora_targets=tuple(ac["target_modules"]),
                lora_dropout=float(ac.get("lora_dropout", 0.0)))
    return lora, state


def save_peft_adapter(model, adapter_dir: str):
    """what `model.llm.save_pretrained(dir)` writes under peft 0.6.0: the adapter directory `peft_ckpt` reloads"""
    import json
    os.makedirs(adapter_dir, exist_ok=True)
    cfg = model.cfg
    ac = dict(peft_type="LORA", task_type="CAUSAL_LM", base_model_name_or_path=None, r=cfg["lora_r"], lora_alpha=cfg["lora_alpha"],
              lora_dropout=cfg.get("lora_dropout", 0.0), target_modules=list(cfg["lora_targets"]), bias="none", fan_in_fan_out=False,
              inference_mode=False, init_lora_weights=True, layers_pattern=None, layers_to_transform=None, modules_to_save=None,
              revision=None, auto_mapping=None)
    with open(os.path.join(adapter_dir, "adapter_config.json"), "w") as f:
        json.dump(ac, f, indent=2)
    sd = {}
    for k, v in model.state_dict().items():
        if k.startswith(PEFT_PREFIX) and (".lora_A." in k or ".lora_B." in k):
            sd[k[len(PEFT_PREFIX):].replace(".default.weight", ".weight")] = v.detach().cpu().clone()
    torch.save(sd, os.path.join(adapter_dir, "adapter_model.bin"))
    return sorted(sd)


def _load_state(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu")


def model_factory(train_config, model_config, **kwargs):
    """returns (model, tokenizer) like src/slam_llm/models/slam_model.py:21-51.

    `train_config.enable_ddp` (the flag the reference's pipeline reads before `DDP(model)`, pipeline/finetune.py:181-184)
    puts the module in autograd_params mode: its trainable parameters then receive their gradients through autograd, so
    torch's DistributedDataParallel reducer works on it unmodified; `++model_config.autograd_params=false` keeps the
    flat-buffer backward for callers that drive `slam_llm_amd.train.GradSync` themselves."""
    check_supported(train_config, model_config)
    cfg = build_config(train_config, model_config)
    peft_state = None
    if kwargs.get("peft_ckpt", None):   # slam_model.py:210-213: the adapter directory defines the LoRA, use_peft or not
        logger.info("loading peft_ckpt from: %s", kwargs.get("peft_ckpt"))
        lora, peft_state = read_peft_adapter(str(kwargs.get("peft_ckpt")))
        cfg.update(lora)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("slam_llm_amd.model_factory: no HIP device visible (the HIP path has no CPU fallback)")
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)   # ops launch on the CURRENT device's current stream (reference: finetune.py:128-131)
    dev = torch.device("cuda", local_rank)
    tokenizer = None
    llm_path = _get(model_config, "llm_path", None)
    if llm_path and os.path.isdir(str(llm_path)):
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(llm_path)
        tokenizer.pad_token_id = tokenizer.eos_token_id  # slam_model.py:63-64
    if "autograd_params" not in kwargs:
        kwargs["autograd_params"] = bool(_get(model_config, "autograd_params", _get(train_config, "enable_ddp", False)))
    seed = int(_get(train_config, "seed", 42))
    model = SlamHipModel(cfg, dev, tokenizer=tokenizer, train_config=train_config, model_config=model_config, **kwargs)
    if _get(model_config, "random_init", False):
        model.init_random(seed)
    else:
        W = {}
        for key in ("encoder_state", "llm_state"):
            p = _get(model_config, key, None)
            if p:
                st = _load_state(str(p))
                if key == "encoder_state" and isinstance(st, dict) and "cfg" in st and "model" in st:
                    # a raw checkpoint {"cfg": ..., "model": state_dict}: WavLM's as models/encoder.py:118-121 reads it (flat cfg, the module
                    # sits under `encoder.model.`), or fairseq's HuBERT as load_model_ensemble_and_task reads it (models/encoder.py:130-140:
                    # the module values are nested under cfg["model"], and the fairseq model IS `self.encoder`: prefix `encoder.`).  Its cfg
                    # carries the dropout / layerdrop values the reference builds the module with; explicit recipe values win.
                    enc_kind = cfg.get("encoder_name")
                    ck_cfg = st["cfg"]
                    if enc_kind == "hubert" and isinstance(ck_cfg, dict) and isinstance(ck_cfg.get("model"), dict):
                        ck_cfg = ck_cfg["model"]
                    for mine, theirs in (("hub_dropout", "dropout"), ("hub_attention_dropout", "attention_dropout"),
                                         ("hub_activation_dropout", "activation_dropout"), ("hub_dropout_input", "dropout_input"),
                                         ("hub_layerdrop", "encoder_layerdrop")):
                        if theirs in ck_cfg and _get(model_config, "encoder_" + theirs.replace("encoder_", ""), None) is None:
                            cfg[mine] = float(ck_cfg[theirs])
                    cfg["hub_regularisers_from"] = "checkpoint cfg"
                    prefix = "encoder." if enc_kind == "hubert" else "encoder.model."
                    st = {prefix + k: v for k, v in st["model"].items()}
                W.update(st)
        if not W:
            raise FileNotFoundError("no weights given: set model_config.encoder_state / llm_state (state dicts in the "
                                    "reference's key names) or model_config.random_init=true")
        model.load_weights(W, seed=seed)   # projector / LoRA tensors absent from W get the reference's fresh-module init
    if cfg.get("encoder_name") in ("hubert", "wavlm") and not cfg.get("freeze_encoder", True):
        # the reference builds the un-frozen module from its checkpoint's own cfg (the published Large checkpoints carry 0.0 for several of
        # these); a plain state dict has no cfg, so say which values the train-mode regularisers run with
        logger.warning("un-frozen %s encoder: train-mode regularisers from %s: dropout=%s attention_dropout=%s activation_dropout=%s "
                       "dropout_input=%s encoder_layerdrop=%s (override with ++model_config.encoder_dropout=... etc.)", cfg["encoder_name"],
                       cfg.get("hub_regularisers_from", "the module defaults / recipe (no checkpoint cfg seen)"), cfg.get("hub_dropout"),
                       cfg.get("hub_attention_dropout"), cfg.get("hub_activation_dropout"), cfg.get("hub_dropout_input"), cfg.get("hub_layerdrop"))
    if peft_state is not None:
        own = set(model.state_dict().keys())
        unknown = [k for k in peft_state if k not in own]
        missing = [k for k in own if (".lora_A." in k or ".lora_B." in k) and k not in peft_state]
        if unknown or missing:
            raise RuntimeError(f"peft_ckpt does not match the model: {len(unknown)} unknown keys (e.g. {unknown[:2]}), "
                               f"{len(missing)} adapter tensors missing (e.g. {missing[:2]})")
        model.load_state_dict(peft_state, strict=False)
        model.mark_params_updated()
    ckpt_path = kwargs.get("ckpt_path", None)  # projector/LoRA checkpoint written by save_model_checkpoint_peft
    if ckpt_path is not None:
        logger.info("loading other parts from: %s", ckpt_path)
        model.load_state_dict(torch.load(ckpt_path, map_location="cpu"), strict=False)
        model.mark_params_updated()
    return model, tokenizer


def get_speech_dataset(dataset_config, tokenizer, split):
    """dataset plugin entry: kaldi-ark multitask layout (aispeech_asr) or JSONL (asr_librispeech), chosen from the config"""
    from slam_llm_amd.dataset import get_speech_dataset as _get
    return _get(dataset_config, tokenizer, split)


def inference_batch(model, tokenizer, dataloader, decode_log: str, device="cuda", **generate_kwargs):
    """Loop body of the reference's batch decoder (src/slam_llm/pipeline/inference_batch.py:118-137): move the batch,
    `model.generate(**batch)`, `tokenizer.batch_decode(..., skip_special_tokens=True)`, write `<decode_log>_pred` /
    `<decode_log>_gt` as "key\ttext" lines.  Returns the number of utterances written."""
    import torch
    n = 0
    with open(decode_log + "_pred", "w") as pred, open(decode_log + "_gt", "w") as gt:
        for batch in dataloader:
            batch = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
            tokens = model.generate(**batch, **generate_kwargs)
            texts = tokenizer.batch_decode(tokens, add_special_tokens=False, skip_special_tokens=True)
            for key, text, target in zip(batch["keys"], texts, batch["targets"]):
                pred.write(f"{key}\t{text.replace(chr(10), ' ')}\n")
                gt.write(f"{key}\t{target}\n")
                n += 1
    return n
