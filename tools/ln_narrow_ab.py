"""GPU (tools): LayerNorm at the Whisper shape (46500 x 1280) -- one row per wave (SLAM_LN_NARROW=0) against 2 / 4 rows per wave; run once per setting (the knob is read once).
python tools/ln_narrow_ab.py  (prints us, TB/s and a checksum of the output bits)"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_llm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for M2, d2 in ((46500, 1280), (12000, 512), (46500, 1024)):
    g = torch.Generator(device=dev).manual_seed(3)
    x2 = torch.randn(M2, d2, generator=g, device=dev).to(torch.bfloat16)
    w2 = torch.randn(d2, generator=g, device=dev)
    b2 = torch.randn(d2, generator=g, device=dev)
    y = ops.layernorm(x2, w2, b2, 1e-5)
    torch.cuda.synchronize()
    h = hashlib.sha1(y.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            ops.layernorm(x2, w2, b2, 1e-5)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 20 * 1e3)
    print(f"SLAM_LN_NARROW={os.environ.get('SLAM_LN_NARROW', '2')} {M2}x{d2}: {best:6.1f} us {2 * M2 * d2 * 2 / best / 1e6:.2f} TB/s  bits {h}")
