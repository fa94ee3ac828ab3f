"""GPU: the SURVEY 8d RAGGED set (clip durations ~U(2 s, 30 s), seed 1235, pad_or_trim off, answers ~U{8..128}) through the
dynamic-frame batcher, four ways on the SAME clips:
  padded          reference semantics: zero-padded mel batch through the encoder, padded [B, T_max] LLM pass
  packed_llm      ++model_config.varlen=true: pad rows dropped before the LLM (seg_lo/seg_hi attention, per-token RoPE)
  ragged          + ++model_config.varlen_encoder=true: per-clip frame counts through the encoder (no pad frames)
  ragged_sum36k   ragged, batches formed by the packed-aware budget (sum of real tokens <= 36 000 instead of B * T_max <= 12 000)
  ragged_sum_hbm  ragged, packed-aware budget = batcher.frames_for_hbm() (~61 k real tokens, ~250 clips per batch: the buffer size the
                  288 GB part is for -- north_star "ragged buffers sized for 288 GB"; its own 900-clip draw of the same distribution)
  c5              BASELINE configs[4] style (aispeech_asr multi-task ASR + ST): the same clips with DYNAMIC PROMPTS (one of several task
                  prompts per sample, 6 .. 40 tokens), the recipe's default LoRA (r 64, alpha 16, all seven projections, dropout 0.05),
                  ragged encoder + packed LLM, dynamic-frame batcher
Prints audio-seconds/sec (true clip durations, SURVEY 8d metric) for each, and one bench-style JSON line per variant whose
config.workload names it."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from slam_llm_amd import batcher
    from slam_llm_amd.model import SlamAdamW, SlamHipModel, make_config
    from slam_llm_amd.train import train_step
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1235)
    n_clips = 320
    secs = torch.rand(n_clips, generator=g) * 28 + 2
    samples = []
    for s_ in secs.tolist():
        n = int(s_ * 16000) // 160 * 160
        alen = batcher.whisper_audio_length(n, 5, pad_to_30s=False)
        A = int(torch.randint(8, 129, (1,), generator=g))
        samples.append(batcher.make_sample(torch.zeros(n), torch.randint(3, 128000, (16,), generator=g).tolist(),
                                           torch.randint(3, 128000, (A - 1,), generator=g).tolist(), 2, alen))
    # C5: a multi-task stream -- each sample draws one of eight task prompts (ASR / translation into several languages, with and
    # without a hot-word list): prompt lengths 6 .. 40 tokens, translation answers ~1.3x as long
    task_len = [6, 9, 12, 17, 22, 28, 34, 40]
    samples_c5 = []
    for s_, smp in zip(secs.tolist(), samples):
        n = int(s_ * 16000) // 160 * 160
        alen = batcher.whisper_audio_length(n, 5, pad_to_30s=False)
        task = int(torch.randint(0, 8, (1,), generator=g))
        A = int(torch.randint(8, 129, (1,), generator=g))
        A = int(A * (1.3 if task >= 4 else 1.0))
        samples_c5.append(batcher.make_sample(torch.zeros(n), torch.randint(3, 128000, (task_len[task],), generator=g).tolist(),
                                              torch.randint(3, 128000, (A - 1,), generator=g).tolist(), 2, alen))
    # the 288 GB-sized row needs more clips than the 320 above (one batch holds ~250): its own draw of the same distribution
    g2 = torch.Generator().manual_seed(1237)
    samples_hbm = []
    for s_ in (torch.rand(900, generator=g2) * 28 + 2).tolist():
        n = int(s_ * 16000) // 160 * 160
        A = int(torch.randint(8, 129, (1,), generator=g2))
        samples_hbm.append(batcher.make_sample(torch.zeros(n), torch.randint(3, 128000, (16,), generator=g2).tolist(),
                                               torch.randint(3, 128000, (A - 1,), generator=g2).tolist(), 2,
                                               batcher.whisper_audio_length(n, 5, pad_to_30s=False)))
    hbm_budget = batcher.frames_for_hbm()
    res = {}
    lora7 = dict(lora_r=64, lora_alpha=16, lora_targets=("q_proj", "k_proj", "v_proj", "o_proj", "up_proj", "gate_proj", "down_proj"))
    variants = [("padded", dict(), 12000, "padded"), ("packed_llm", dict(varlen=True), 12000, "padded"),
                ("ragged", dict(varlen=True, varlen_encoder=True), 12000, "padded"),
                ("ragged_sum36k", dict(varlen=True, varlen_encoder=True), 36000, "sum"),
                ("ragged_sum_hbm", dict(varlen=True, varlen_encoder=True), hbm_budget, "sum"),
                ("c5", dict(varlen=True, varlen_encoder=True, **lora7), 12000, "padded")]
    describe = {"padded": "padded (reference semantics: zero-padded mel batch, padded [B, T_max] LLM pass)",
                "packed_llm": "model_config.varlen=true (pad rows dropped before the LLM)",
                "ragged": "varlen + varlen_encoder (no pad frames through the encoder, packed LLM pass)",
                "ragged_sum36k": "varlen + varlen_encoder, dataset_config.frame_budget=sum, train_max_frame_length=36000",
                "ragged_sum_hbm": f"varlen + varlen_encoder, dataset_config.frame_budget=sum, train_max_frame_length=frames_for_hbm()={hbm_budget} "
                                  f"(the 288 GB-sized ragged buffer)",
                "c5": "C5 style: multi-task dynamic prompts (6..40 tokens), LoRA r64 alpha16 on all seven projections, varlen + varlen_encoder"}
    only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
    for name, extra, mfl, budget in variants:
        if only and name not in only:
            continue
        pool = samples_c5 if name == "c5" else (samples_hbm if name == "ragged_sum_hbm" else samples)
        groups = list(batcher.dynamic_batches(iter(pool), mfl, budget=budget))[:-1]
        if budget == "padded":
            groups = groups[:5]
        else:
            groups = groups[:3]
        kw = dict(lora_r=16, lora_alpha=32, lora_targets=("q_proj", "v_proj"), lora_dropout=0.05, pad_or_trim=False)
        kw.update(extra)
        cfg = make_config("whisper-large-v3", "llama-3-8b", **kw)
        model = SlamHipModel(cfg, dev).init_random(42)
        model.train()
        opt = SlamAdamW(model, lr=1e-4)
        batches = []
        for grp in groups:
            b = batcher.collate(grp, 0, left_pad_prompt=False)
            b["audio"] = (torch.randn(b["audio"].shape, generator=g) * 0.1).clamp_(-1, 1)
            batches.append({k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()})
        audio_s = sum(float(b["audio_len"].sum()) / 16000 for b in batches)
        tokens = sum(int(b["attention_mask"].sum()) for b in batches)
        padded = sum(b["attention_mask"].numel() for b in batches)
        enc_rows_padded = sum(b["audio"].shape[0] * ((b["audio"].shape[1] // 160 + 1) // 2) for b in batches)
        enc_rows_real = sum(sum((n // 160 + 1) // 2 for n in b["audio_len_list"]) for b in batches)
        for b in batches:        # one untimed pass over every batch shape (allocator, RoPE tables, kernel caches warm)
            train_step(model, b, opt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches:
            train_step(model, b, opt)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = dict(audio_s_per_s=round(audio_s / dt, 1), ms_per_batch=round(dt / len(batches) * 1e3, 1), batches=len(batches),
                         clips=sum(len(g_) for g_ in groups), llm_tokens_valid=tokens, llm_tokens_padded=padded,
                         encoder_rows_real=enc_rows_real, encoder_rows_padded=enc_rows_padded,
                         peak_hbm_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1))
        print(name, res[name], flush=True)
        print(json.dumps({"metric": "audio-seconds/sec (true clip durations; ragged set, not the headline workload)",
                          "value": res[name]["audio_s_per_s"], "unit": "audio-s/s", "n_gpus": 1, "ms_per_step": res[name]["ms_per_batch"],
                          "steps": len(batches), "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": f"RAGGED C3 set (clip durations ~U(2 s, 30 s), seed 1235, pad_or_trim off, answers ~U{{8..128}}, "
                                                 f"dynamic-frame batcher): {describe[name]}; whisper-large-v3 -> llama-3-8b, full optimizer "
                                                 f"step, GPU log-mel in the step", **{k: res[name][k] for k in ("batches", "clips", "llm_tokens_valid",
                                                 "llm_tokens_padded", "encoder_rows_real", "encoder_rows_padded", "peak_hbm_gb")}}}), flush=True)
        del model, opt, batches
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
