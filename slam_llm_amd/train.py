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


def setup_distributed(device_type: str = "cuda", init_single: bool = False):
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, local_rank, world).
    The device is selected BEFORE the process group exists and handed to it as `device_id` (RCCL then builds its communicator
    eagerly on that device instead of guessing one at the first collective).  `init_single` also initialises the group at
    world size 1 (tests: the RCCL code path on a 1-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or init_single) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # "nccl" IS RCCL on ROCm; SLAM_DIST_BACKEND=gloo lets two ranks share one GPU for functional tests
        backend = os.environ.get("SLAM_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        kw = {}
        if device_type == "cuda":
            idx = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(idx)
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", idx)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def rccl_version() -> Optional[str]:
    """version of the collective library behind backend "nccl" (RCCL on ROCm), e.g. "2.26.6"; None when unavailable"""
    try:
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        return None


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

    def __init__(self, source, bucket_bytes: int = 32 << 20, group=None, force_collectives: bool = False):
        self.source = source
        self.bucket = max(1, bucket_bytes // 4)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # world size 1 normally skips the collectives (nothing to average); force_collectives launches them anyway so that the
        # RCCL path (ReduceOp selection, async handles on flat-buffer views, stream ordering) runs on a 1-GPU box
        self.force = bool(force_collectives) and dist.is_initialized()
        self.launched = 0          # collectives launched so far (tests / bench report)
        self.exposed_ms = None     # bench: HIP-event time of the last finish() (the part of the exchange the backward did not hide)
        self.time_finish = False
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
        if (self.world == 1 and not self.force) or not self.armed:
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
        self.launched += 1
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
        ev = None
        if self.time_finish and self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if (self.world > 1 or self.force) and self.armed and self.done < self.flat.numel():
            self.on_prefix(self.flat.numel())
        self._wait()
        if ev is not None:
            ev[1].record()
            self._finish_events = getattr(self, "_finish_events", [])
            self._finish_events.append(ev)

    def exposed_ms_per_step(self) -> Optional[float]:
        """mean HIP-event time of finish() over the calls timed so far (time_finish=True): tail launch + waits on the compute
        stream = the exposed part of the gradient exchange.  Synchronises."""
        evs = getattr(self, "_finish_events", [])
        if not evs:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / len(evs)


def all_ranks_have_data(has_batch: bool, device) -> bool:
    """Uneven-input guard (reference: DDP `Join`, utils/train_utils.py:91; DeepSpeed path: gloo monitored_barrier,
    utils/deepspeed_utils.py:110-131).  Policy: the epoch ends for everybody as soon as one rank runs dry."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return has_batch
    flag = torch.tensor([1 if has_batch else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def train_step(model, batch: dict, optimizer, scheduler=None, grad_sync: Optional[GradSync] = None,
               gradient_accumulation_steps: int = 1, do_step: bool = True, scaler=None):
    """One iteration of train_utils.py:112-169.  Returns (loss, acc) as device tensors (no host sync).
    With gradient accumulation the caller passes do_step=False on all but the last micro-step (the reference's
    `(step + 1) % gradient_accumulation_steps == 0` test, train_utils.py:132/153).
    `scaler` (a torch.cuda.amp.GradScaler) selects the reference's `use_fp16` branch (train_utils.py:70-76,128-150):
    forward under `torch.cuda.amp.autocast`, `scaler.scale(loss).backward()`, `scaler.step(optimizer)`, `scaler.update()`.
    The HIP path computes in bf16 with fp32 masters whatever the autocast state says; the scale factor reaches the kernels as the
    backward's incoming gradient (a power of two: exact in bf16 and fp32) and GradScaler unscales the flat gradient views in place."""
    if grad_sync is not None:
        grad_sync.arm(do_step)   # one reduction of the accumulated buffer instead of one per micro-step (same result)
    if scaler is not None:
        with torch.autocast("cuda", dtype=torch.float16):
            outputs, acc = model(**batch)
    else:
        outputs, acc = model(**batch)
    loss = outputs.loss
    if gradient_accumulation_steps != 1:
        loss = loss / gradient_accumulation_steps
    if scaler is not None:
        scaler.scale(loss).backward()
    else:
        loss.backward()
    if do_step:
        if grad_sync is not None:
            grad_sync.finish()
        if scaler is not None:
            scaler.step(optimizer)
            scaler.update()
        else:
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
