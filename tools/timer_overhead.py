"""GPU: what the per-launch HIP events of bench.py's KernelTimer cost the C3 step they measure.  One process, one model, the timed region of
bench.py (`measure`) run alternately with `ops.TIMER` set (every GEMM / attention / LoRA-hop launch bracketed by two events: ~1 300 events per
step) and unset; 3 rounds x 6 steps each.  Prints one JSON object."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402
from slam_llm_amd import ops  # noqa: E402
from slam_llm_amd.model import SlamAdamW, SlamHipModel  # noqa: E402
from slam_llm_amd.slam_model_hip import build_config  # noqa: E402
from slam_llm_amd.train import lr_lambda, train_step  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    wl = bench.WORKLOADS["c3"]
    cfg = build_config(dict(use_peft=True, peft_config=wl["peft"], seed=42, freeze_encoder=True), wl["model"])
    model = SlamHipModel(cfg, dev).init_random(42)
    model.train()
    opt = SlamAdamW(model, lr=1e-4, weight_decay=0.0)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: lr_lambda(s, 1000, 100000))
    batch, T, Ta = bench.make_batch(cfg, wl["clips"], dev, seed=1234)

    def step():
        return train_step(model, batch, opt, sched, None)

    bench.measure(step, 0, 3, 1, dist, torch.cuda.synchronize, dev)
    res = {"timer": [], "plain": []}
    mode_filter = os.environ.get("SLAM_TIMER_FILTER")
    for rnd in range(3):
        for mode in ("timer", "plain"):
            ops.TIMER = ops.KernelTimer() if mode == "timer" else None
            _, elapsed, _ = bench.measure(step, 6, 0, 1, dist, torch.cuda.synchronize, dev)
            n_events = sum(len(v) for v in ops.TIMER.rec.values()) * 2 / 6 if ops.TIMER is not None else 0
            ops.TIMER = None
            res[mode].append(elapsed / 6 * 1e3)
            print(mode, round(elapsed / 6 * 1e3, 2), "ms/step", int(n_events), "events/step", file=sys.stderr, flush=True)
    print(json.dumps({k: {"ms_per_step": [round(x, 2) for x in v], "median": round(statistics.median(v), 2)} for k, v in res.items()}))


if __name__ == "__main__":
    main()
