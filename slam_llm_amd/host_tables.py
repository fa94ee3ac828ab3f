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
