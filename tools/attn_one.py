"""GPU: time the Whisper-large-v3 encoder attention forward of the C3 batch (B=31, T=1500, 20 heads, D=64, no mask)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_llm_amd import ops
dev = torch.device("cuda:0")
B, T, H, D = 31, 1500, 20, 64
qkv = torch.randn(B * T, 3 * H * D, device=dev).to(torch.bfloat16)
vt = ops.head_rope_transpose(qkv, 2 * H * D, B, T, H, D)
o = torch.empty(B * T, H * D, device=dev, dtype=torch.bfloat16)
f = lambda: ops.attn_fwd(qkv[:, :H * D], qkv[:, H * D:2 * H * D], vt, B, T, H, H, D, False, D ** -0.5, want_lse=False, out=o)
for _ in range(3):
    f()
torch.cuda.synchronize()
best = 1e9
for rnd in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        f()
    e.record()
    torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) * 100)
print(f"whisper attn fwd: {best:.1f} us  {4.0 * B * H * T * T * D / best / 1e6:.1f} TF")
from slam_llm_amd.lib import call
call("slam_attn_set_fwd_qf", 1)
for _ in range(3):
    f()
torch.cuda.synchronize()
best = 1e9
for rnd in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        f()
    e.record()
    torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) * 100)
print(f"whisper attn fwd QF=1: {best:.1f} us  {4.0 * B * H * T * T * D / best / 1e6:.1f} TF")
