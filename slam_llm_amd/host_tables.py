"""Small host-side constant tables (built once with torch on CPU, uploaded to HBM by the caller)."""
import math

import torch


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """openai-whisper's fixed encoder positional embedding (model.py `sinusoids`), [length, channels] fp32."""
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([t.sin(), t.cos()], dim=1)


def rope_tables(T: int, D: int, theta: float):
    """HF LlamaRotaryEmbedding (default rope): cos/sin of pos * theta^(-2i/D), [T, D/2] fp32 each;
    positions are arange(T) for every row because the reference never passes position_ids."""
    inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.arange(T, dtype=torch.float32)[:, None] * inv[None, :]
    return fr.cos().contiguous(), fr.sin().contiguous()


def wavlm_relative_buckets(T: int, num_buckets: int, max_distance: int) -> torch.Tensor:
    """bucket index of every relative distance k - q in [-(T-1), T-1] (entry i <-> distance i - (T-1)), WavLM's bidirectional
    T5-style bucketing (src/slam_llm/models/wavlm/modules.py:417-442): one half of the buckets per sign, distances below
    num_buckets/4 exact, beyond that log-spaced up to max_distance and clamped to the last bucket.  int64 [2T-1]."""
    import math
    rel = torch.arange(-(T - 1), T)
    half = num_buckets // 2
    exact = half // 2
    a = rel.abs()
    log_part = exact + (torch.log(a.float() / exact) / math.log(max_distance / exact) * (half - exact)).to(torch.long)
    log_part = torch.clamp(log_part, max=half - 1)
    return (rel > 0).to(torch.long) * half + torch.where(a < exact, a, log_part)
