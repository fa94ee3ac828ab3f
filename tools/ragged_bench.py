"""GPU: the SURVEY 8d RAGGED set (clip durations ~U(2 s, 30 s), seed 1235, pad_or_trim off, answers ~U{8..128}) through the
reference's dynamic-frame batcher (B * T_max <= 12000, right-padding collator) -- padded LLM pass vs cfg["varlen"] (packed
sequences, no pad tokens).  Prints audio-seconds/sec for both on the SAME batches (true clip durations, SURVEY 8d metric)."""
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
    n_clips = 160
    secs = torch.rand(n_clips, generator=g) * 28 + 2
    samples = []
    for s_ in secs.tolist():
        n = int(s_ * 16000) // 160 * 160
        alen = batcher.whisper_audio_length(n, 5, pad_to_30s=False)
        A = int(torch.randint(8, 129, (1,), generator=g))
        samples.append(batcher.make_sample(torch.zeros(n), torch.randint(3, 128000, (16,), generator=g).tolist(),
                                           torch.randint(3, 128000, (A - 1,), generator=g).tolist(), 2, alen))
    groups = list(batcher.dynamic_batches(iter(samples), 12000))[:-1]
    res = {}
    for varlen in (False, True):
        cfg = make_config("whisper-large-v3", "llama-3-8b", lora_r=16, lora_alpha=32, lora_targets=("q_proj", "v_proj"),
                          lora_dropout=0.05, pad_or_trim=False, varlen=varlen)
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
        for b in batches:        # one untimed pass over every batch shape (allocator, RoPE tables, kernel caches warm)
            train_step(model, b, opt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches:
            train_step(model, b, opt)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res["packed" if varlen else "padded"] = dict(audio_s_per_s=audio_s / dt, ms_per_batch=dt / len(batches) * 1e3)
        res["batches"], res["clips"], res["valid_tokens"], res["padded_tokens"] = len(batches), sum(len(g_) for g_ in groups), tokens, padded
        del model, opt, batches
        torch.cuda.empty_cache()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
