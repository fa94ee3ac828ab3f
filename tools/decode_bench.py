"""GPU: decode throughput of SlamHipModel.generate at the C3 dims (Whisper-large-v3 -> Llama-3-8B, LoRA r16),
the reference's inference_batch setting: beam 4, do_sample False (scripts/decode_*.sh).  Prints one JSON line.

HBM roofline of one decode step: every frozen weight matrix of the LLM is read once (bf16), plus the live KV cache."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--beams", type=int, default=4)
    ap.add_argument("--new", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--encoder", default="whisper-large-v3")
    ap.add_argument("--llm", default="llama-3-8b")
    ap.add_argument("--splits", type=int, default=0, help="override the skinny-GEMM cross-workgroup split plan")
    ap.add_argument("--kernels", action="store_true", help="per-kernel HIP-event timing (adds launch gaps)")
    args = ap.parse_args()
    from slam_llm_amd import ops
    from slam_llm_amd.model import SlamHipModel, make_config
    dev = torch.device("cuda:0")
    ops.SKINNY_SPLITS = args.splits
    cfg = make_config(args.encoder, args.llm, lora_r=16, lora_alpha=32, lora_targets=("q_proj", "v_proj"))
    model = SlamHipModel(cfg, dev).init_random(42)
    model.eval()
    B = args.batch
    g = torch.Generator().manual_seed(7)
    audio = (torch.randn(B, int(args.seconds * 16000), generator=g) * 0.1).clamp(-1, 1).to(dev)
    Ta = (int(args.seconds * 100) // 2 + (int(args.seconds * 100) % 2)) // cfg["ds_rate"] if args.seconds < 30 else 1500 // cfg["ds_rate"]
    P = 20
    T = Ta + P
    ids = torch.cat([torch.zeros(B, Ta, dtype=torch.int64), torch.randint(3, cfg["vocab"], (B, P), generator=g)], dim=1).to(dev)
    mm = torch.zeros(B, T, dtype=torch.bool, device=dev)
    mm[:, :Ta] = True
    batch = dict(input_ids=ids, attention_mask=torch.ones(B, T, dtype=torch.bool, device=dev), audio=audio, modality_mask=mm)
    kw = dict(max_new_tokens=args.new, num_beams=args.beams, eos_token_id=cfg["vocab"] - 1, pad_token_id=0, min_length=args.new)
    def run(n_new):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.generate(**{k: v.clone() for k, v in batch.items()}, **dict(kw, max_new_tokens=n_new, min_length=n_new))
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out

    run(4)  # warm-up
    # marginal cost of a decode step = slope between two generation lengths (fixed costs -- encoder, prefill, graph
    # capture -- cancel); the fixed part is reported separately
    n1 = 16
    t1, _ = run(n1)
    if args.kernels:
        ops.TIMER = ops.KernelTimer()
    t_all, out = run(n1 + args.new)
    timer, ops.TIMER = ops.TIMER, None
    if args.kernels:
        ops.TIMER = ops.KernelTimer()
        t1, _ = run(n1)
        ops.TIMER = None
    ms_step = (t_all - t1) / args.new * 1e3
    t_prefill = t1 - n1 * ms_step * 1e-3
    d, Hq, Hkv, D, Fd, V, Ln = (cfg["llm_dim"], cfg["llm_heads"], cfg["llm_kv_heads"], cfg["llm_head_dim"], cfg["llm_ffn"],
                                 cfg["vocab"], cfg["llm_layers"])
    w_bytes = 2 * (Ln * (d * (Hq + 2 * Hkv) * D + Hq * D * d + 3 * d * Fd) + V * d)
    R = B * args.beams
    kv_bytes = 2 * 2 * Ln * (B * T + R * (n1 + args.new // 2)) * Hkv * D  # prompt KV is shared by the beams of an item
    res = {"metric": "decode tokens/sec (beam hypotheses advance together)", "rows": R, "batch": B, "beams": args.beams,
           "prompt_len": T, "new_tokens": int(out.shape[1]), "fixed_ms (encoder + prefill + graph capture)": t_prefill * 1e3, "ms_per_decode_step": ms_step,
           "emitted_tokens_per_s": B * 1e3 / ms_step, "hypothesis_tokens_per_s": R * 1e3 / ms_step,
           "roofline": {"bound": "hbm", "algorithmic_bytes_per_step": w_bytes + kv_bytes, "weights_bytes": w_bytes,
                        "kv_bytes": kv_bytes, "achieved": (w_bytes + kv_bytes) / (ms_step * 1e-3) / 1e9, "peak": 8000.0,
                        "unit": "GB/s", "frac": (w_bytes + kv_bytes) / (ms_step * 1e-3) / 1e9 / 8000.0}}
    if timer is not None:
        ks = timer.summary()
        res["kernels"] = {k: dict(launches=v["launches"], total_ms=v["total_ms"], avg_ms=v["avg_ms"]) for k, v in ks.items()}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
