"""Training-step driver and data-parallel gradient exchange for the HIP path.

`train_step` is the loop body of the reference's `train()` (src/slam_llm/utils/train_utils.py:112-169, the
non-fp16 branch): forward, `loss / gradient_accumulation_steps`, backward, optimizer + scheduler step,
zero_grad -- without the per-step host syncs the reference forces (`.tolist()` at slam_model.py:384, the
f-string of loss/acc at train_utils.py:171): loss/acc stay on the device until the caller asks for them.

Data parallelism (reference: `DDP(model)` at src/slam_llm/pipeline/finetune.py:181-184): one process per GPU,
all weights replicated, ONE collective per optimizer step -- a mean all-reduce over the flat trainable-gradient
buffer (RCCL over xGMI; 113 MB fp32 for Whisper-large-v3 -> Llama-3-8B r16).  The buffer is laid out in
backward-production order, so `GradSync` launches the all-reduce bucket by bucket on prefixes while the
remaining LLM backward is still running; only the projector tail is exposed.  Loss is a per-rank token mean and
gradients are averaged over ranks (mean of means), exactly like the reference under DDP.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def setup_distributed(device_type: str = "cuda"):
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" IS RCCL on ROCm; SLAM_DIST_BACKEND=gloo lets two ranks share one GPU for functional tests
        backend = os.environ.get("SLAM_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        if device_type == "cuda":
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class GradSync:
    """Bucketed mean all-reduce of the flat fp32 gradient buffer, launched on buffer prefixes as they complete.

    Semantics follow DDP (src/slam_llm/pipeline/finetune.py:181-184): every ARMED backward ends with the gradient
    buffer averaged over ranks.  Two rules make gradient accumulation correct:
      * `on_backward_begin` (called by the model before its first backward kernel) waits for every collective the
        previous backward left in flight and resets the prefix cursor, so a backward never accumulates into memory a
        collective is still reading/writing and every backward reduces its own prefixes (DDP without `no_sync`:
        avg(avg(g1) + g2) = avg(g1) + avg(g2));
      * `train_step` disarms the object on micro-steps that do not end in an optimizer step: the sum of micro-step
        gradients is linear, so ONE reduction of the accumulated buffer on the last micro-step gives the same result
        with 1/k of the traffic.
    `source` is the model (its `store.grad` is read at launch time, so a re-allocated buffer is followed) or a tensor."""

    def __init__(self, source, bucket_bytes: int = 32 << 20, group=None):
        self.source = source
        self.bucket = max(1, bucket_bytes // 4)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.done = 0
        self.armed = True
        self.handles = []
        # ncclAvg needs (R)CCL >= 2.10; otherwise (and on gloo) sum and scale after the wait
        self.avg_native = False
        if dist.is_initialized() and dist.get_backend(group) == "nccl":
            try:
                self.avg_native = tuple(torch.cuda.nccl.version()[:2]) >= (2, 10)
            except Exception:  # noqa: BLE001  (version query unavailable: stay on the always-valid SUM path)
                self.avg_native = False

    @property
    def flat(self) -> torch.Tensor:
        src = self.source
        return src if isinstance(src, torch.Tensor) else src.store.grad

    def attach(self, model):
        if getattr(model, "autograd_params", False):
            raise RuntimeError("GradSync is the fast path of the flat-buffer backward; a model in autograd_params mode is "
                               "reduced by torch's DistributedDataParallel instead")
        model.grad_hooks.append(self)
        return self

    def arm(self, on: bool = True):
        self.armed = bool(on)
        return self

    def on_backward_begin(self):
        """a new backward is about to write the gradient buffer: retire whatever the previous one left in flight"""
        if self.handles or self.done:
            self._wait()

    def on_prefix(self, end: int):
        """gradients in flat[0:end] are final for this backward"""
        if self.world == 1 or not self.armed:
            return
        flat = self.flat
        total = flat.numel()
        if end - self.done < self.bucket and end < total:
            return
        if end <= self.done:
            return
        view = flat[self.done:end]
        op = dist.ReduceOp.AVG if self.avg_native else dist.ReduceOp.SUM
        h = dist.all_reduce(view, op=op, group=self.group, async_op=True)
        self.handles.append((h, view))
        self.done = end

    __call__ = on_prefix   # plain-callable hook form

    def _wait(self):
        for h, view in self.handles:
            h.wait()
            if not self.avg_native:
                view.div_(self.world)
        self.handles.clear()
        self.done = 0

    def finish(self):
        """flush the tail and wait for all buckets (call before optimizer.step())."""
        if self.world > 1 and self.armed and self.done < self.flat.numel():
            self.on_prefix(self.flat.numel())
        self._wait()


def all_ranks_have_data(has_batch: bool, device) -> bool:
    """Uneven-input guard (reference: DDP `Join`, utils/train_utils.py:91; DeepSpeed path: gloo monitored_barrier,
    utils/deepspeed_utils.py:110-131).  Policy: the epoch ends for everybody as soon as one rank runs dry."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return has_batch
    flag = torch.tensor([1 if has_batch else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def train_step(model, batch: dict, optimizer, scheduler=None, grad_sync: Optional[GradSync] = None,
               gradient_accumulation_steps: int = 1, do_step: bool = True):
    """One iteration of train_utils.py:112-169.  Returns (loss, acc) as device tensors (no host sync).
    With gradient accumulation the caller passes do_step=False on all but the last micro-step (the reference's
    `(step + 1) % gradient_accumulation_steps == 0` test, train_utils.py:132/153)."""
    if grad_sync is not None:
        grad_sync.arm(do_step)   # one reduction of the accumulated buffer instead of one per micro-step (same result)
    outputs, acc = model(**batch)
    loss = outputs.loss
    if gradient_accumulation_steps != 1:
        loss = loss / gradient_accumulation_steps
    loss.backward()
    if do_step:
        if grad_sync is not None:
            grad_sync.finish()
        optimizer.step()
        if scheduler is not None:
            scheduler.step()
        optimizer.zero_grad()
    return loss.detach(), acc


def lr_lambda(step: int, warmup_steps: int, total_steps: int) -> float:
    """linear warm-up then linear decay to 0 (src/slam_llm/pipeline/finetune.py:253-260)."""
    if step < warmup_steps:
        return min(step / warmup_steps, 1)
    return max(0.0, 1 - (step - warmup_steps) / (total_steps - warmup_steps))
