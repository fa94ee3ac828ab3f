"""TEST INFRASTRUCTURE -- runs the REFERENCE's AnyPrecisionAdamW (src/slam_llm/policies/anyprecision_optimizer.py:16-178, loaded by
path, unmodified; it only needs torch) on a seeded parameter / gradient sequence and stores parameters and states after every
step -> tests/golden/anyprecision.npz.  Cases: the configuration pipeline/finetune.py:237-245 builds (bf16 momentum + variance,
no Kahan) on fp32 parameters (this build keeps fp32 masters) and on bf16 parameters (the reference's pure_bf16 route, g8), and
the Kahan-compensated form on bf16 parameters."""
import importlib.util
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "anyprecision.npz")
SRC = "/root/reference/src/slam_llm/policies/anyprecision_optimizer.py"

CASES = {   # name: (param dtype, kahan, weight_decay)
    "fp32_params": (torch.float32, False, 0.01),
    "bf16_params": (torch.bfloat16, False, 0.01),
    "bf16_params_kahan": (torch.bfloat16, True, 0.0),
}
N, STEPS, LR = 4096, 6, 3e-3


def inputs():
    g = torch.Generator().manual_seed(77)
    p0 = torch.randn(N, generator=g) * 0.5
    grads = [torch.randn(N, generator=g) * (0.1 if s % 2 else 1.0) * torch.logspace(-3, 0, N) for s in range(STEPS)]
    return p0, grads


def main():
    spec = importlib.util.spec_from_file_location("ref_anyprecision", SRC)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    p0, grads = inputs()
    fx = {"p0": p0.numpy(), "lr": np.float64(LR)}
    for s, gr in enumerate(grads):
        fx[f"grad.{s}"] = gr.numpy()
    for name, (pdt, kahan, wd) in CASES.items():
        p = torch.nn.Parameter(p0.to(pdt).clone())
        opt = mod.AnyPrecisionAdamW([p], lr=LR, weight_decay=wd, use_kahan_summation=kahan, momentum_dtype=torch.bfloat16,
                                    variance_dtype=torch.bfloat16)
        for s, gr in enumerate(grads):
            p.grad = gr.to(pdt).clone()
            opt.step()
            st = opt.state[p]
            fx[f"{name}.p.{s}"] = p.detach().float().numpy().copy()
            fx[f"{name}.m.{s}"] = st["exp_avg"].float().numpy().copy()
            fx[f"{name}.v.{s}"] = st["exp_avg_sq"].float().numpy().copy()
            if kahan:
                fx[f"{name}.c.{s}"] = st["compensation"].float().numpy().copy()
    np.savez_compressed(OUT, **fx)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
