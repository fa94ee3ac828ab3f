"""GPU: where do torch.optim.AdamW(model.parameters()) and the fused SlamAdamW part ways on the real (tiny) model?  One step each from
the same weights and batch; prints the largest parameter differences with the gradient / moment values at those elements."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from oracle import slam_oracle as O  # noqa: E402  (tools may use the oracle's synthetic batches)
from slam_llm_amd.model import SlamAdamW, SlamHipModel  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(O.make_config(), lora_dropout=0.0)
W = O.init_weights(cfg, seed=42)
audio = O.synth_audio(2, 1.0, seed=300)
ob = O.synth_batch(cfg, audio, prompt_len=5, answer_lens=(4, 7), seed=400, left_pad=True, pad_to_30s=False)
b = {k: v.to(dev) for k, v in ob.items()}


def run(kind, steps):
    m = SlamHipModel(dict(cfg), dev).load_weights(W)
    m.train()
    opt = SlamAdamW(m, lr=1e-3) if kind == "slam" else torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.0)
    gs = []
    for _ in range(steps):
        out, _ = m(**{k: v.clone() for k, v in b.items()})
        out.loss.backward()
        gs.append(m.store.grad.clone())
        opt.step()
        opt.zero_grad()
    return m, gs


for steps in (1, 2, 3):
    ms, gs = run("slam", steps)
    mt, gt = run("torch", steps)
    d = (ms.store.flat - mt.store.flat).abs()
    print(f"steps {steps}: max |dp| {float(d.max()):.3e}  max|p| {float(ms.store.flat.abs().max()):.3e}  grads equal: "
          f"{[bool(torch.equal(a, c)) for a, c in zip(gs, gt)]}")
    idx = torch.topk(d, 5).indices.tolist()
    offs = sorted(((off, n) for n, (off, cnt, _) in ms.store.offsets.items()))
    for i in idx:
        name = [n for off, n in offs if off <= i][-1]
        print(f"   flat[{i}] ({name}): slam {float(ms.store.flat[i]):+.6e} torch {float(mt.store.flat[i]):+.6e} init {float(SlamHipModel(dict(cfg), dev).load_weights(W).store.flat[i]):+.6e} "
              f"grads " + " ".join(f"{float(g[i]):+.3e}" for g in gs))
