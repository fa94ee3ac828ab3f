"""GPU: HuBERT-large conv front end at TRUE dimensions (BASELINE config 4: 6 x 30 s raw clips), stage by stage with HIP events:
im2col + MFMA GEMM + fused LayerNorm-GELU per conv layer, feature projection, grouped positional conv.  Prints algorithmic HBM
bytes (input read once + output written once per stage, SURVEY 8d rule for bandwidth-bound kernels), time and GB/s per stage."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from slam_llm_amd import ops
    from slam_llm_amd.model import HipHubertEncoder
    from slam_llm_amd.slam_model_hip import HUBERT_PRESETS
    dev = torch.device("cuda:0")
    cfg = dict(HUBERT_PRESETS["hubert-large"])
    enc = HipHubertEncoder(cfg, dev).init_random(42)
    w = enc.w
    B, N = 6, 480000
    wav = torch.nn.functional.layer_norm(torch.randn(B, N, device=dev) * 0.1, (N,))
    rows = []

    def timed(name, nbytes, flops, fn, reps=5):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            out = fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        rows.append(dict(stage=name, MB=round(nbytes / 1e6, 1), us=round(ms * 1e3, 1), GBps=round(nbytes / ms / 1e6, 1),
                         TFLOPs=round(flops / ms / 1e9, 1) if flops else None))
        return out

    x2d, Tin, cin = wav.contiguous().view(B * N, 1), N, 1
    for i, (co, k, st) in enumerate(zip(cfg["hub_conv_dim"], cfg["hub_conv_kernel"], cfg["hub_conv_stride"])):
        Kp = w[f"c{i}"].shape[1]
        Tout = (Tin - k) // st + 1
        in_b = x2d.numel() * x2d.element_size()
        cols, _ = timed(f"conv{i} im2col (k{k} s{st}, {cin}->{Kp} cols)", in_b + B * Tout * Kp * 2, 0,
                        lambda: ops.conv1d_im2col(x2d, B, Tin, 0, cin, k, st, 0, Kp=Kp))
        y = timed(f"conv{i} GEMM [{B * Tout} x {co} x {Kp}]", B * Tout * (Kp + co) * 2 + co * Kp * 2, 2.0 * B * Tout * co * k * cin,
                  lambda: ops.gemm_nt(cols, w[f"c{i}"], bias=w[f"c{i}_b"]))
        x2d = timed(f"conv{i} LayerNorm+GELU [{B * Tout} x {co}]", 2 * B * Tout * co * 2, 0,
                    lambda: ops.layernorm(y, w[f"c{i}_lw"], w[f"c{i}_lb"], 1e-5, gelu=True))
        del cols, y
        Tin, cin = Tout, co
    T, d = Tin, cfg["hub_dim"]
    M = B * T
    h = timed("feature-projection LayerNorm", 2 * M * cin * 2, 0, lambda: ops.layernorm(x2d, w["fp_lw"], w["fp_lb"], cfg["hub_eps"]))
    h = timed(f"feature-projection GEMM [{M} x {d} x {cin}]", M * (cin + d) * 2 + d * cin * 2, 2.0 * M * d * cin,
              lambda: ops.gemm_nt(h, w["fp"], bias=w["fp_b"]))
    G, kpos = cfg["hub_pos_groups"], cfg["hub_pos_k"]
    gch = d // G
    x = torch.empty((M, d), dtype=torch.bfloat16, device=dev)
    cols = torch.empty((M, enc.pos_kp), dtype=torch.bfloat16, device=dev)

    def posconv():
        for g in range(G):
            ops.conv1d_im2col(h, B, T, g * gch, gch, kpos, 1, kpos // 2, Kp=enc.pos_kp, Tout_limit=T, out=cols)
            ops.gemm_nt(cols, w["pos"][g], out=x[:, g * gch:(g + 1) * gch], bias=w["pos_b"][g * gch:(g + 1) * gch],
                        act=ops.ACT_GELU, residual=h[:, g * gch:(g + 1) * gch])
    timed(f"positional conv (k{kpos}, {G} groups: {G} x [im2col + GEMM {M} x {gch} x {enc.pos_kp}])", 3 * M * d * 2 + d * kpos * gch * 2,
          2.0 * M * d * kpos * gch, posconv)
    tot = sum(r["us"] for r in rows)
    print(json.dumps(dict(rows=rows, total_us=round(tot, 1), frames=T, clips=B), indent=1))


if __name__ == "__main__":
    main()
