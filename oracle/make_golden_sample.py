"""Fixture for SAMPLING decode (do_sample=True): the reference's slam_model.generate (src/slam_llm/models/slam_model.py:409-456,
UNMODIFIED, imported from /root/reference) -> HF generate on CPU with torch.manual_seed(seed), against the oracle restatement
seeded the same way (CPU multinomial is deterministic given the generator state, so tokens must agree exactly).

HF's GenerationConfig top_k default changed across the versions the reference allows (transformers>=4.31: 50 in 4.x, unset in
5.x); the fixture sets llm.generation_config.top_k explicitly so that it does not depend on the installed version.

Run in the build container: python oracle/make_golden_sample.py -> tests/golden/generate_sample.npz"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import slam_oracle as O  # noqa: E402
from oracle.make_golden import GOLD, build_reference  # noqa: E402
from oracle.make_golden_cases import GENERATE_CASE  # noqa: E402

# (seed 13 of the third run was dropped: one draw sits on an fp32 knife edge between the oracle's full-sequence recompute and HF's
# KV-cache logits; the tokens split at step 9 of one row)
RUNS = (  # (num_beams, temperature, top_k, top_p, repetition_penalty, seed)
    (1, 1.0, 50, 1.0, 1.0, 11), (1, 0.7, 50, 0.9, 1.0, 12), (1, 1.3, 0, 0.8, 1.2, 23),
    (4, 1.0, 50, 1.0, 1.0, 14), (4, 0.8, 20, 0.95, 1.0, 15), (3, 1.0, 0, 0.9, 1.3, 16))


def key(nb, temp, tk, tp, rp, seed):
    return f"tokens.b{nb}.t{temp}.k{tk}.p{tp}.rp{rp}.seed{seed}"


def main():
    case = GENERATE_CASE
    cfg = case["cfg"]
    audio = O.synth_audio(len(case["clip_samples"]), 2.0, seed=1234)
    batch = O.synth_infer_batch(cfg, audio, case["clip_samples"], case["prompt_lens"])
    scale = 5.0
    W = O.init_weights(cfg, seed=42)
    W["llm.base_model.model.lm_head.weight"] = W["llm.base_model.model.lm_head.weight"] * scale
    model = build_reference(cfg, W)
    model.eval()
    fx = {"scale": np.float64(scale)}
    gold_gen = np.load(os.path.join(GOLD, "generate.npz"))
    eos = int(gold_gen[f"s{scale}.eos"])
    fx["eos"] = np.int64(eos)
    for nb, temp, tk, tp, rp, seed in RUNS:
        model.tokenizer = types.SimpleNamespace(bos_token_id=case["bos"], eos_token_id=eos, pad_token_id=1)
        model.llm.generation_config.top_k = tk if tk else None
        torch.manual_seed(seed)
        with torch.no_grad():
            out = model.generate(**{k: v.clone() for k, v in batch.items()}, max_new_tokens=case["max_new_tokens"], num_beams=nb,
                                 do_sample=True, temperature=temp, top_p=tp, repetition_penalty=rp)
        torch.manual_seed(seed)
        mine = O.slam_generate(W, cfg, {k: v.clone() for k, v in batch.items()}, max_new_tokens=case["max_new_tokens"],
                               num_beams=nb, eos=eos, pad=1, repetition_penalty=rp,
                               sample=dict(temperature=temp, top_k=tk, top_p=tp))
        ok = out.shape == mine.shape and bool((out == mine).all())
        print(f"sample beams={nb} T={temp} top_k={tk} top_p={tp} rp={rp} seed={seed}: oracle match {ok}\n{out.numpy()}")
        if not ok:
            print("oracle:\n", mine.numpy())
        fx[key(nb, temp, tk, tp, rp, seed)] = out.numpy()
    np.savez_compressed(os.path.join(GOLD, "generate_sample.npz"), **fx)
    print("generate_sample.npz written")


if __name__ == "__main__":
    main()
