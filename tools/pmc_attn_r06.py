"""GPU (tools): a few launches of the attention kernels of the C3 step at HEAD for rocprofv3 --pmc passes (tools/profile_r06.sh): Whisper forward
on accumulators started at -m (Q pre-scaled: the shipped form) and with the general softmax (knob 60), Llama forward, Llama backward (dQ + dK / dV
transposed-read ring kernels).  python tools/pmc_attn_r06.py [n_launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402
from slam_llm_amd.host_tables import rope_tables  # noqa: E402
from slam_llm_amd.lib import call  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
B, T, H, D = 31, 1500, 20, 64
qkv = torch.randn(B * T, 3 * H * D, device=dev).to(torch.bfloat16)
qkv[:, : H * D] = (qkv[:, : H * D].float() * ops.qscale(D ** -0.5)).to(torch.bfloat16)
q2, k2, v2 = qkv[:, : H * D], qkv[:, H * D: 2 * H * D], qkv[:, 2 * H * D:]
o = torch.empty(B * T, H * D, device=dev, dtype=torch.bfloat16)
for knob in (61, 60):
    call("slam_attn_set_fwd_qf", knob)
    for _ in range(n):
        ops.attn_fwd(q2, k2, v2, B, T, H, H, D, False, D ** -0.5, want_lse=False, out=o, q_prescaled=True)
    torch.cuda.synchronize()
call("slam_attn_set_fwd_qf", 61)
Bl, Tl, Hq, Hkv, Dl = 31, 380, 32, 8, 128
qkvl = torch.randn(Bl * Tl, (Hq + 2 * Hkv) * Dl, device=dev).to(torch.bfloat16)
ql, kl, vl = qkvl[:, : Hq * Dl], qkvl[:, Hq * Dl:(Hq + Hkv) * Dl], qkvl[:, (Hq + Hkv) * Dl:]
km = torch.zeros((Bl, ops.round_up(Tl, 64)), dtype=torch.uint8, device=dev)
km[:, :Tl] = 1
cos, sin = (t.to(dev) for t in rope_tables(Tl, Dl, 500000.0))
ol, lsel = ops.attn_fwd(ql, kl, vl, Bl, Tl, Hq, Hkv, Dl, True, Dl ** -0.5, key_mask=km)
dol = torch.randn(Bl * Tl, Hq * Dl, device=dev).to(torch.bfloat16)
dqkv = torch.empty_like(qkvl)
for _ in range(n):
    ops.attn_fwd(ql, kl, vl, Bl, Tl, Hq, Hkv, Dl, True, Dl ** -0.5, key_mask=km, out=ol)
for _ in range(n):
    ops.attn_bwd(ql, kl, vl, ol, dol, lsel, dqkv[:, : Hq * Dl], dqkv[:, Hq * Dl:(Hq + Hkv) * Dl], dqkv[:, (Hq + Hkv) * Dl:],
                 Bl, Tl, Hq, Hkv, Dl, True, Dl ** -0.5, key_mask=km, rope=(cos, sin))
torch.cuda.synchronize()
