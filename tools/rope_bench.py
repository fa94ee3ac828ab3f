"""GPU: head_rope_transpose variants at the LLM shape of the C3 batch (B=31, T=380, 32 q / 8 kv heads, D=128)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_llm_amd import ops
from slam_llm_amd.host_tables import rope_tables
dev = torch.device("cuda:0")
B, T, Hq, Hkv, D = 31, 380, 32, 8, 128
qkv = torch.randn(B * T, (Hq + 2 * Hkv) * D, device=dev).to(torch.bfloat16)
cos, sin = (t.to(dev) for t in rope_tables(T, D, 500000.0))
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
MB = B * T * Hq * D * 2 / 1e6
for name, f, traffic in (
    ("q rope + transpose", lambda: ops.head_rope_transpose(qkv, 0, B, T, Hq, D, cos=cos, sin=sin), 3 * MB),
    ("q rope only (in place)", lambda: ops.head_rope_transpose(qkv, 0, B, T, Hq, D, cos=cos, sin=sin, want_t=False), 2 * MB),
    ("q transpose only", lambda: ops.head_rope_transpose(qkv, 0, B, T, Hq, D), 2 * MB),
    ("k rope + transpose", lambda: ops.head_rope_transpose(qkv, Hq * D, B, T, Hkv, D, cos=cos, sin=sin), 3 * MB / 4),
    ("v transpose only", lambda: ops.head_rope_transpose(qkv, (Hq + Hkv) * D, B, T, Hkv, D), 2 * MB / 4)):
    us = t(f)
    print(f"{name:26s} {us:7.1f} us  {traffic / us:.2f} TB/s")
