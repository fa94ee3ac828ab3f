"""GPU: where a wave of the dQ kernel spends a 32-key tile (cycle stamps of the probe form, slam_attn_set_bwd_variant(14)).

Stamps per tile: 0 loop top, 1 after the vmcnt wait + barrier, 2 after the DMA issue, 3 after the S / dP products were issued,
4 after the softmax, 5 after the dQ (dV / dK) products were issued.  dQ: workgroup (2, 5, 3) = queries 256-383 of one head (12 tiles),
waves 0 / 3; `python tools/attn_dq_probe.py dkdv`: the dK/dV ring kernel, workgroup (0, 3, 5) = keys 0-127, waves 0 / 7, first 12 tiles.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402
from slam_llm_amd.lib import call  # noqa: E402

dev = torch.device("cuda:0")
B, T, Hq, Hkv, D = 31, 380, 32, 8, 128
qkv = torch.randn(B * T, (Hq + 2 * Hkv) * D, device=dev).to(torch.bfloat16)
q2, k2, v2 = qkv[:, : Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
qt = ops.head_rope_transpose(qkv, 0, B, T, Hq, D)
kt = ops.head_rope_transpose(qkv, Hq * D, B, T, Hkv, D)
vt = ops.head_rope_transpose(qkv, (Hq + Hkv) * D, B, T, Hkv, D)
Tp = vt.shape[-1]
km = torch.zeros((B, Tp), dtype=torch.uint8, device=dev)
km[:, :T] = 1
scale = D ** -0.5
o, lse = ops.attn_fwd(q2, k2, vt, B, T, Hq, Hkv, D, True, scale, key_mask=km)
do = torch.randn(B * T, Hq * D, device=dev).to(torch.bfloat16)
dot = ops.head_rope_transpose(do, 0, B, T, Hq, D)
dqkv = torch.empty_like(qkv)
WHICH = sys.argv[1] if len(sys.argv) > 1 else "dq"   # dq | dkdv
call("slam_attn_set_bwd_variant", 14 if WHICH == "dq" else 19)
for _ in range(5):
    ops.attn_bwd(q2, k2, v2, o, do, lse, dqkv[:, : Hq * D], dqkv[:, Hq * D:(Hq + Hkv) * D], dqkv[:, (Hq + Hkv) * D:],
                 B, T, Hq, Hkv, D, True, scale, key_mask=km)
torch.cuda.synchronize()
call("slam_attn_set_bwd_variant", 0)
out = np.zeros(256, dtype=np.uint64)
call("slam_attn_debug_clock", out.ctypes.data)
st = out.reshape(2, 16, 8).astype(np.int64)
names = ["wait+barrier", "DMA issue", "reads + S/dP", "softmax", "dQ products" if WHICH == "dq" else "dV/dK products", "-> next top"]
for w, wn in enumerate(("wave 0", "wave 3" if WHICH == "dq" else "wave 7")):
    print(wn)
    for it in range(12):
        r = st[w, it]
        d = [r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], (st[w, it + 1][0] - r[5]) if it < 11 else 0]
        print(f"  tile {it:2d}: " + "  ".join(f"{n} {int(v):5d}" for n, v in zip(names, d)) + f"   total {int(r[5] - r[0])}")
