"""Measures what tests/test_model_gpu.py bounds for the un-frozen WavLM in train mode, over several mask draws (VERDICT r4 next #1a).

For cases A..E of tests/golden/wavlm_train_tiny.npz and N torch seeds (the counter-based masks derive from torch.initial_seed()): the
error of the small gate gradients (grep_linear.bias [8], grep_a [H]) as a fraction of the layer's grep_linear.weight gradient norm --
the quantity `SMALL_GATE_BOUND` bounds -- and the worst cosine / norm deviation of every other gradient.  Writes a markdown table.

    python tools/wavlm_trainmode_seeds.py [--seeds 5] [--out gpurun_out/r05_wavlm_trainmode_seeds.md]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_wavlm_trainmode_seeds.md"))
    a = ap.parse_args()
    from tests import test_model_gpu as T
    dev = torch.device("cuda:0")
    lines = ["| case | seed | worst small-gate err / ‖d grep_linear.weight‖ (which) | worst cosine (which) | worst norm deviation | outside the test's bounds |",
             "|---|---|---|---|---|---|"]
    worst_all = 0.0
    for tag in "ABCDE":
        for seed in [None] + list(range(101, 101 + a.seeds)):
            rep = {}
            T._wavlm_train_mode_case(dev, tag, mask_seed=seed if seed is not None else 20240924, report=rep)
            sg = rep.get("small_gate_err_over_weight_grad_norm", {})
            wname, wval = max(sg.items(), key=lambda kv: kv[1]) if sg else ("-", 0.0)
            worst_all = max(worst_all, wval)
            cn = rep.get("cos_norm", {})
            cname, (cval, _) = min(cn.items(), key=lambda kv: kv[1][0]) if cn else ("-", (1.0, 0.0))
            nworst = max((v[1] for v in cn.values()), default=0.0)
            short = lambda n: n.replace("encoder.model.encoder.", "").replace("encoder.model.", "")      # noqa: E731
            lines.append(f"| {tag} | {'suite' if seed is None else seed} | {wval:.4f} ({short(wname)}) | {cval:.5f} ({short(cname)}) | {nworst:.4f} | {rep['bad']} |")
    lines.append("")
    lines.append(f"worst small-gate ratio over all rows: **{worst_all:.4f}**; tests/test_model_gpu.py `SMALL_GATE_BOUND` = {T.SMALL_GATE_BOUND}")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
