"""GPU: where a wave of the forward attention kernel spends a 64-key tile at the Whisper shape (cycle stamps of the probe form).

Stamps per tile: 0 loop top, 1 after the first barrier, 2 after the LDS stores + second barrier, 3 after the next tile's global
loads were issued, 4 after the S products were issued, 5 after the softmax, 6 after the PV products were issued.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402
from slam_llm_amd.lib import call  # noqa: E402

dev = torch.device("cuda:0")
B, T, H, D = 31, 1500, 20, 64
qkv = torch.randn(B * T, 3 * H * D, device=dev).to(torch.bfloat16)
vt = ops.head_rope_transpose(qkv, 2 * H * D, B, T, H, D)
d = H * D
obuf = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
scale = D ** -0.5


def fwd():
    ops.attn_fwd(qkv[:, :d], qkv[:, d: 2 * d], vt, B, T, H, H, D, False, scale, want_lse=False, out=obuf)


def timed(n=10):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fwd()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


# forms timed interleaved (the clocks follow the recent load): (fragments per wave, LDS-DMA tiles)
FORMS = [("2 fragments per wave, DMA tiles, mask-free instantiation: 3 waves / SIMD (shipped)", 2, 11, 31),
         ("2 fragments per wave, DMA tiles, general instantiation: 2 waves / SIMD", 2, 11, 30), ("1 fragment per wave, DMA tiles", 1, 11, 30),
         ("2 fragments per wave, register-staged tiles", 2, 10, 30), ("1 fragment per wave, register-staged tiles", 1, 10, 30)]
res = {n: [] for n, _, _, _ in FORMS}
for rnd in range(4):
    for n, qf, dma, plain in FORMS:
        call("slam_attn_set_fwd_qf", qf)
        call("slam_attn_set_fwd_qf", dma)
        call("slam_attn_set_fwd_qf", plain)
        fwd()
        res[n].append(timed())
call("slam_attn_set_fwd_qf", 0)
call("slam_attn_set_fwd_qf", 11)
call("slam_attn_set_fwd_qf", 31)
for n, _, _, _ in FORMS:
    v = sorted(res[n][1:])
    print(f"fwd, {n}: median {v[1]:.1f} us  {4.0 * B * H * T * T * D / v[1] / 1e6:.1f} TF")
call("slam_attn_set_bwd_variant", 14)
for _ in range(3):
    fwd()
torch.cuda.synchronize()
call("slam_attn_set_bwd_variant", 0)
out = np.zeros(256, dtype=np.uint64)
call("slam_attn_debug_clock", out.ctypes.data)
st = out.reshape(2, 16, 8).astype(np.int64)
names = ["barrier 1", "lds store + barrier 2", "gload issue", "K reads + S", "softmax", "PV", "-> next top"]
for w, wn in enumerate(("wave 0", "wave 3")):
    print(wn)
    for it in range(15):
        r = st[w, it]
        dd = [r[i + 1] - r[i] for i in range(6)] + [st[w, it + 1][0] - r[6]]
        print(f"  tile {it:2d}: " + "  ".join(f"{n} {int(v):5d}" for n, v in zip(names, dd)) + f"   total {int(st[w, it + 1][0] - r[0])}")
