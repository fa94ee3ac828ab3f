"""GPU (tools): does HIP_FORCE_DEV_KERNARG still take effect when it is set AFTER `import torch` but before the first HIP call?  Times 20 000 back-to-back launches of a
tiny kernel (the launch path is what the knob changes).  python tools/kernarg_probe.py <before|after|unset>"""
import os
import sys
import time

mode = sys.argv[1]
os.environ.pop("HIP_FORCE_DEV_KERNARG", None)
if mode == "before":
    os.environ["HIP_FORCE_DEV_KERNARG"] = "1"
import torch  # noqa: E402

if mode == "after":
    os.environ["HIP_FORCE_DEV_KERNARG"] = "1"
x = torch.zeros(64, device="cuda:0")
for _ in range(2000):
    x.add_(1.0)
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(20000):
        x.add_(1.0)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20000 * 1e6)
print(f"{mode}: {best:.2f} us per launch")
