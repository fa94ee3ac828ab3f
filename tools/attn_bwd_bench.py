"""GPU: time attention fwd / bwd at the LLM shape of the C3 workload (B=31, T=380, 32 q / 8 kv heads, D=128, causal)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402
from slam_llm_amd.host_tables import rope_tables  # noqa: E402

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
cos, sin = (t.to(dev) for t in rope_tables(T, D, 500000.0))
scale = D ** -0.5
o, lse = ops.attn_fwd(q2, k2, vt, B, T, Hq, Hkv, D, True, scale, key_mask=km)
do = torch.randn(B * T, Hq * D, device=dev).to(torch.bfloat16)
dot = ops.head_rope_transpose(do, 0, B, T, Hq, D)
dqkv = torch.empty_like(qkv)


def bwd():
    ops.attn_bwd(q2, k2, v2, o, do, lse, dqkv[:, : Hq * D], dqkv[:, Hq * D:(Hq + Hkv) * D],
                 dqkv[:, (Hq + Hkv) * D:], B, T, Hq, Hkv, D, True, scale, key_mask=km, rope=(cos, sin))


def fwd():
    ops.attn_fwd(q2, k2, vt, B, T, Hq, Hkv, D, True, scale, key_mask=km, out=o)


from slam_llm_amd.lib import call as _call  # noqa: E402


def variant(v):
    def run():
        _call("slam_attn_set_bwd_variant", v)
        bwd()
        _call("slam_attn_set_bwd_variant", 0)
    return run


bwd_old = variant(1)
ABL = {11: "no DMA in loop", 12: "no softmax VALU", 15: "no barrier"}


def timed(f, n=10):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


# the clocks follow the recent load (power limit): variants are timed INTERLEAVED, five rounds, best and median reported
CASES = [("fwd", fwd, 4.0), ("bwd (ring kernels)", bwd, 10.0), ("bwd (round-1 kernels)", bwd_old, 10.0),
         ("bwd (ring dK/dV, register-staged dQ)", variant(2), 10.0)] + [(f"bwd ablation: {n}", variant(v), 10.0) for v, n in ABL.items()]
for _, f, _ in CASES:
    for _ in range(3):
        f()
res = {n: [] for n, _, _ in CASES}
for _ in range(5):
    for n, f, _ in CASES:
        res[n].append(timed(f))
for n, _, c in CASES:
    v = sorted(res[n])
    flops = c * B * Hq * T * T * D * 0.5
    print(f"{n}: best {v[0]:.1f} us  median {v[2]:.1f} us  {flops / v[2] / 1e6:.1f} TF", flush=True)

from slam_llm_amd.lib import call  # noqa: E402
# forward forms, interleaved: query fragments per wave x tile staging (11 = LDS-DMA ring, shipped)
FWD = [("2 fragments per wave, DMA tiles (shipped)", 2, 11), ("1 fragment per wave, DMA tiles", 1, 11),
       ("1 fragment per wave, register-staged tiles", 1, 10), ("2 fragments per wave, register-staged tiles", 2, 10)]
fres = {n: [] for n, _, _ in FWD}
for rnd in range(5):
    for n, qf, dma in FWD:
        call("slam_attn_set_fwd_qf", qf)
        call("slam_attn_set_fwd_qf", dma)
        fwd()
        fres[n].append(timed(fwd))
call("slam_attn_set_fwd_qf", 0)
call("slam_attn_set_fwd_qf", 11)
for n, _, _ in FWD:
    v = sorted(fres[n])
    print(f"fwd, {n}: median {v[2]:.1f} us  {4.0 * B * Hq * T * T * D * 0.5 / v[2] / 1e6:.1f} TF", flush=True)
