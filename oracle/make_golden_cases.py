"""Golden-fixture case table shared by oracle/make_golden.py and the tests -- TEST INFRASTRUCTURE."""
from oracle import slam_oracle as O

CASES = {
    # MHA-ish GQA(2q/1kv), hd 64, left-padded ragged answers (exercises g3/g4 padding semantics)
    "step_tiny": dict(cfg=O.make_config(), clip_seconds=2.0, answer_lens=(5, 9), left_pad=True),
    # hd 128, GQA 4q/2kv, 128 mel bins, LoRA r16 on q,k,v,o, right-pad collator (aispeech_asr layout)
    "step_hd128": dict(cfg=O.make_config(n_mels=128, enc_dim=128, enc_heads=2, llm_dim=256, llm_heads=4, llm_kv_heads=2,
                                         llm_head_dim=128, llm_ffn=512, vocab=1024, rope_theta=500000.0, lora_r=16,
                                         lora_alpha=32, lora_targets=("q_proj", "k_proj", "v_proj", "o_proj")),
                       clip_seconds=3.0, answer_lens=(7, 3, 11), left_pad=False),
}

# HuBERT-style encoder (a11): same structure as hubert-large, narrow widths (conv 64 ch, d 128, pos-conv k 16 / 4 groups)
HUBERT_TINY = O.hubert_config(hub_conv_dim=(64,) * 7, hub_dim=128, hub_heads=2, hub_layers=2, hub_ffn=256, hub_pos_k=16,
                              hub_pos_groups=4)

# Q-Former projector (a3'): the reference module hard-codes Blip2QFormerConfig() widths (768/12/3072); layers, queries,
# encoder_dim and llm_dim come from the recipe -> keep those small
QFORMER_CASE = dict(cfg=O.qformer_config(qf_layers=2, qf_queries=8), enc_dim=128, llm_dim=128, B=2, Tk=37, masked_tail=7)

# generate (f1): step_tiny architecture with a sharpened lm_head: x24 gives clear ranking margins (end-to-end GPU
# parity), x5 a flatter distribution where beam search departs from greedy (host-logic parity); ragged clips and
# prompts -> left padding; eos id is chosen by make_golden so that hypotheses finish at different lengths
GENERATE_CASE = dict(cfg=O.make_config(), lm_head_scales=(24.0, 5.0), clip_samples=(32000, 22400, 28800), prompt_lens=(6, 4, 7),
                     max_new_tokens=12, pad=0, bos=1)

# cov1d-linear projector (f4, models/projector.py:29-49): T not a multiple of k (tail frames dropped by the strided conv)
COV1D_CASE = dict(enc_dim=128, llm_dim=192, k=5, B=2, T=43)

# unfrozen-encoder training (f4, train_config.freeze_encoder=false): step_tiny's architecture, odd mel frame count (conv2's last
# window hangs over the edge) and T2 not a multiple of the projector's k (tail frames get zero gradient)
UNFROZEN_CASE = dict(cfg=O.make_config(), clip_seconds=1.77, answer_lens=(5, 9), left_pad=True, lr=2e-3)

# WavLM encoder (f4, models/wavlm/WavLM.py): WavLM-Large's structure (layer_norm extractor, no conv bias, layer_norm_first, gated
# relative position bias), narrow widths; 40 buckets / max distance 24 so that the log-spaced and the clamped buckets both occur
# inside ~50 frames
WAVLM_TINY = O.wavlm_config(hub_conv_dim=(64,) * 7, hub_dim=128, hub_heads=2, hub_layers=2, hub_ffn=256, hub_pos_k=16,
                            hub_pos_groups=4, wavlm_buckets=40, wavlm_max_distance=24)
# the same module left in TRAIN mode (freeze_encoder=false, slam_model.py:317-318): three layers so that layerdrop can skip one and leave
# two, every regulariser of WavLM.py:180-185 non-zero (the released cfg's activation_dropout / dropout_input are 0; 0.1 here so that
# their placement is pinned too)
WAVLM_TRAIN_TINY = dict(WAVLM_TINY, hub_layers=3)
WAVLM_TRAIN_REG = dict(dropout=0.1, attention_dropout=0.1, activation_dropout=0.1, dropout_input=0.1, encoder_layerdrop=0.4)
# HuBERT-base structure (HF: feat_extract_norm="group", do_stable_layer_norm=False, conv_bias=False) at toy widths
HUBERT_BASE_TINY = O.hubert_base_config(hub_conv_dim=(64,) * 7, hub_dim=128, hub_heads=2, hub_layers=2, hub_ffn=256, hub_pos_k=16,
                                        hub_pos_groups=4)
# WavLM Base structure (group-norm extractor, post-LN layers) at toy widths
WAVLM_BASE_TINY = O.wavlm_base_config(hub_conv_dim=(64,) * 7, hub_dim=128, hub_heads=2, hub_layers=2, hub_ffn=256, hub_pos_k=16,
                                      hub_pos_groups=4, wavlm_buckets=40, wavlm_max_distance=24)
