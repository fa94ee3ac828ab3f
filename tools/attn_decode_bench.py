"""GPU: time slam_attn_decode alone (Llama-3-8B head geometry) over prompt length / generated length / LoRA on-off."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402
from slam_llm_amd.host_tables import rope_tables  # noqa: E402

dev = torch.device("cuda:0")
Hq, Hkv, D, G, beams = 32, 8, 128, 200, 4
NQ = (Hq + 2 * Hkv) * D
for B in (4, 16):
    R = B * beams
    for Tp in (16, 320, 1280):
        for n in (0, 100):
            for lora in (0, 32):
                qkv = torch.randn(R, NQ + lora, device=dev).to(torch.bfloat16)
                lb = torch.randn(NQ, 64, device=dev).to(torch.bfloat16) if lora else None
                Kp = torch.randn(B, Tp, Hkv * D, device=dev).to(torch.bfloat16)
                Vp = torch.randn_like(Kp)
                Kg = torch.randn(R, G, Hkv * D, device=dev).to(torch.bfloat16)
                Vg = torch.randn_like(Kg)
                anc = torch.arange(R, dtype=torch.int32, device=dev)[:, None].repeat(1, G).contiguous()
                start = torch.zeros(B, dtype=torch.int32, device=dev)
                pos = torch.full((R,), Tp + n, dtype=torch.int32, device=dev)
                cos, sin = rope_tables(Tp + G, D, 500000.0)
                cos, sin = cos.to(dev), sin.to(dev)
                out = torch.empty(R, Hq * D, device=dev, dtype=torch.bfloat16)
                f = lambda: ops.attn_decode(qkv, lb, lora, cos, sin, pos, Kp, Vp, start, Kg, Vg, anc, None, n, beams, Hq, Hkv,  # noqa: E731
                                            D, D ** -0.5, out)
                for _ in range(5):
                    f()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(200):
                    f()
                e.record()
                torch.cuda.synchronize()
                us = s.elapsed_time(e) * 1e3 / 200
                kvb = 2 * 2 * (B * Tp + R * (n + 1)) * Hkv * D
                print(f"R={R} Tp={Tp} n={n} lora={lora}: {us:.1f} us  ({kvb / us / 1e3:.0f} GB/s of unique KV)", flush=True)
