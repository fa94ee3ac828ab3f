"""Fixture for UNFROZEN-encoder training (train_config.freeze_encoder=false, src/slam_llm/models/slam_model.py:110-113): the
reference's slam_model (imported UNMODIFIED from /root/reference, built exactly like oracle/make_golden.py's step cases) with the
encoder's parameters left trainable -- three AdamW steps over encoder + projector + LoRA, first-step gradients of EVERY trainable
tensor, losses, final parameters.  Run in the build container: python oracle/make_golden_unfrozen.py ->
tests/golden/step_unfrozen.npz"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import slam_oracle as O  # noqa: E402
from oracle.make_golden import GOLD, build_reference, pack, ref_trainables, wsum  # noqa: E402
from oracle.make_golden_cases import UNFROZEN_CASE  # noqa: E402

# reference module path of each encoder parameter (the adapter in make_golden.py exposes HF WhisperEncoder submodules under
# openai-whisper's names; q/k/v/out and the MLP keep HF's attribute names inside a block)
HF_NAMES = {"attn.query": "self_attn.q_proj", "attn.key": "self_attn.k_proj", "attn.value": "self_attn.v_proj",
            "attn.out": "self_attn.out_proj", "attn_ln": "self_attn_layer_norm", "mlp.0": "fc1", "mlp.2": "fc2",
            "mlp_ln": "final_layer_norm"}


def encoder_params(model, W):
    out = {}
    for n in W:
        if not n.startswith("encoder.") or n.endswith("positional_embedding"):
            continue
        parts = n[len("encoder."):].split(".")
        if parts[0] == "blocks":
            blk = model.encoder.blocks[int(parts[1])].l
            key = ".".join(parts[2:-1])
            mod = blk
            for a in HF_NAMES[key].split("."):
                mod = getattr(mod, a)
        else:
            mod = getattr(model.encoder, parts[0])
        out[n] = getattr(mod, parts[-1])
    return out


def main():
    case = UNFROZEN_CASE
    cfg = case["cfg"]
    torch.manual_seed(0)
    W = O.init_weights(cfg, seed=42)
    model = build_reference(cfg, W)
    tr = dict(ref_trainables(model, cfg))
    tr.update(encoder_params(model, W))
    for n, p_ in tr.items():
        assert tuple(p_.shape) == tuple(W[n].shape) and torch.equal(p_.detach(), W[n]), n
        p_.requires_grad = True
    model.encoder.train()          # what setup_encoder leaves when freeze_encoder is false (no dropout in the Whisper encoder)
    model.train_config.freeze_encoder = False
    audio = O.synth_audio(len(case["answer_lens"]), case["clip_seconds"], seed=1234)
    batch = O.synth_batch(cfg, audio, prompt_len=6, answer_lens=case["answer_lens"], seed=1236, left_pad=case["left_pad"],
                          pad_to_30s=False)
    opt = torch.optim.AdamW(list(tr.values()), lr=case["lr"], weight_decay=0.01)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: O.lr_lambda(s, 2, 10))
    fx = {"weights_sha256": np.array(wsum(W)), "audio": audio.numpy()}
    for k, v in batch.items():
        fx["batch." + k] = v.numpy()
    for step in range(3):
        outputs, acc = model(**{k: v.clone() for k, v in batch.items()})
        outputs.loss.backward()
        if step == 0:
            for n, p_ in tr.items():
                pack(fx, "grad." + n, p_.grad.detach().numpy())
        fx[f"loss.{step}"] = np.float32(outputs.loss.item())
        fx[f"acc.{step}"] = np.float32(float(acc))
        opt.step(); sched.step(); opt.zero_grad()
        print(f"step_unfrozen step {step}: loss {outputs.loss.item():.6f} acc {float(acc):.4f}")
    for n, p_ in tr.items():
        pack(fx, "final." + n, p_.detach().numpy())
    # the oracle restatement on the same inputs
    W2 = O.init_weights(cfg, seed=42)
    outs = O.train_steps(W2, cfg, [dict(batch) for _ in range(3)], lr=case["lr"], weight_decay=0.01, warmup=2, total=10,
                         train_encoder=True)
    for s_ in range(3):
        print("oracle loss", float(outs[s_]["loss"]), "reference", float(fx[f"loss.{s_}"]))
    np.savez_compressed(os.path.join(GOLD, "step_unfrozen.npz"), **fx)
    print("step_unfrozen.npz written", len(tr), "trainable tensors")


if __name__ == "__main__":
    main()
