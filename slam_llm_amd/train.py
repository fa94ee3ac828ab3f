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
        from . import trace
        trace.push("allreduce")
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
        trace.pop()

    def shadow_backward(self, prefix_ends):
        """`Join` for the fast path (reference: `with Join([model])`, utils/train_utils.py:91; torch's default
        `divide_by_initial_world_size=True`): a rank whose shard is exhausted stands in for one armed backward by posting the SAME
        sequence of prefix all-reduces over a ZERO gradient buffer, so the ranks that still have data finish their step instead of
        hanging in a collective nobody else joins.  `prefix_ends` = the prefix sequence a real backward announces
        (`SlamHipModel.prefix_plan()`: the layers' LoRA prefixes, last layer first, then the whole buffer).  The buffer then holds the
        mean over ALL ranks of the active ranks' gradients (the divisor stays the world size, like DDP's Join), so the shadowing rank
        can apply the same optimizer step and its replica stays in sync.  Follow with finish()."""
        self.on_backward_begin()
        self.flat.zero_()
        for end in prefix_ends:
            self.on_prefix(int(end))

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


def ranks_with_data(has_batch: bool, device) -> int:
    """how many ranks still hold a batch this iteration (one SUM all-reduce of a flag).  With it a loop can follow either policy for
    uneven shards: stop as soon as the number drops below the world size (`all_ranks_have_data`, this build's default), or keep going
    until it is 0 with the exhausted ranks shadowing (`train_step(model, None, ...)`, the reference's DDP `Join`, train_utils.py:91)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 1 if has_batch else 0
    flag = torch.tensor([1 if has_batch else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.SUM)
    return int(flag.item())


def train_step(model, batch: Optional[dict], optimizer, scheduler=None, grad_sync: Optional[GradSync] = None,
               gradient_accumulation_steps: int = 1, do_step: bool = True, scaler=None):
    """One iteration of train_utils.py:112-169.  Returns (loss, acc) as device tensors (no host sync).
    With gradient accumulation the caller passes do_step=False on all but the last micro-step (the reference's
    `(step + 1) % gradient_accumulation_steps == 0` test, train_utils.py:132/153).
    `scaler` (a torch.cuda.amp.GradScaler) selects the reference's `use_fp16` branch (train_utils.py:70-76,128-150):
    forward under `torch.cuda.amp.autocast`, `scaler.scale(loss).backward()`, `scaler.step(optimizer)`, `scaler.update()`.
    The HIP path computes in bf16 with fp32 masters whatever the autocast state says; the scale factor reaches the kernels as the
    backward's incoming gradient (a power of two: exact in bf16 and fp32) and GradScaler unscales the flat gradient views in place."""
    if batch is None:
        # this rank's shard is exhausted while others still train (the `Join` policy, see GradSync.shadow_backward): zero gradients
        # through the same collectives, then the same optimizer step on the averaged buffer -- the replicas stay identical
        if grad_sync is None or scaler is not None or gradient_accumulation_steps != 1 or not do_step:
            raise ValueError("train_step(batch=None) shadows one plain optimizer step of the GradSync fast path (no GradScaler, no accumulation)")
        grad_sync.arm(True)
        grad_sync.shadow_backward(model.prefix_plan())
        grad_sync.finish()
        model.attach_grad_views()
        optimizer.step()
        if scheduler is not None:
            scheduler.step()
        optimizer.zero_grad()
        return None, None
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


class GraphedTrainStep:
    """`train_step` captured ONCE in a HIP graph (torch.cuda.CUDAGraph) and replayed: one graph launch instead of the ~1 200 (C3) to
    ~2 300 (C4) kernel launches of the loop body of utils/train_utils.py:112-169.  The launch-bound BASELINE configurations (C1: 69 % of
    the step idle between kernels, C4: ~30 %) are where it pays; the result is bit-identical to the eager step (same kernels, same
    arguments, same order -- tests/test_graph_gpu.py).

    What had to leave the host for this (round 6):
      * the label-row selection runs on the device against a STATIC bound (`label_rows_cap`: HipLlamaLora.label_rows_cap; default = the
        first batch's own count rounded up to a multiple of 64; batches are checked against it one step late, without a sync);
      * the fused AdamW reads lr and the bias corrections from device memory (`slam_adamw_step_dev`): LambdaLR stays a host object,
        its value is written into a 3-float device buffer before every replay;
      * LoRA / Q-Former dropout masks: every dropout-aware kernel XORs a device-resident word into its seed (`slam_set_dropout_salt`),
        and the captured step bumps that word at its end, so each replay draws fresh masks.
    Scope: fixed-shape batches (the padded layouts; `varlen` / `varlen_encoder` batches have data-dependent shapes), SlamAdamW, no
    GradScaler, world size 1 or an un-armed GradSync -- anything else runs the eager `train_step` (same semantics, `self.eager_steps`
    counts them).  The first `warmup` calls also run eagerly (they are real training steps: workspaces, tables and LDS attributes
    of the library are set up by them, outside the capture)."""

    SALT_STEP = 0x9E3779B97F4A7C15 - (1 << 64)      # the golden-ratio increment as a signed int64

    def __init__(self, model, optimizer, scheduler=None, label_rows_cap: Optional[int] = None, warmup: int = 2, grad_sync: Optional[GradSync] = None):
        from .model import SlamAdamW
        if type(optimizer) is not SlamAdamW:
            raise TypeError("GraphedTrainStep captures the fused SlamAdamW step (device-side lr / bias corrections)")
        self.model, self.opt, self.sched, self.gsync = model, optimizer, scheduler, grad_sync
        self.cap_arg, self.warmup = label_rows_cap, int(warmup)
        self.graph = None
        self.key = None
        self.static = None
        self.out = None
        self.calls = self.eager_steps = self.replays = 0
        dev = model.device_
        self.hyper = torch.zeros(3, dtype=torch.float32, device=dev)
        self.hyper_host = torch.zeros(3, dtype=torch.float32).pin_memory()
        self.salt = torch.zeros(1, dtype=torch.int64, device=dev)
        self._count_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._count_ev = None

    # ---- helpers ----
    @staticmethod
    def _signature(batch: dict):
        return tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in batch.items() if isinstance(v, torch.Tensor)))

    def _capturable(self, batch: dict) -> bool:
        m = self.model
        if m.cfg.get("varlen", False) or m.cfg.get("varlen_encoder", False) or m.autograd_params or m.store.pure_bf16:
            return False
        if m.train_encoder:
            return False        # (an un-frozen encoder re-allocates its derived weights on every refresh: not capture-safe, see ops.transpose_into)
        if self.gsync is not None and (self.gsync.world > 1 or self.gsync.force):
            return False        # (collectives stay outside the graph: the eager step overlaps them with the backward)
        if any(not isinstance(v, torch.Tensor) for v in batch.values()):
            return False        # python-side batch entries (length lists of the ragged collators)
        from . import ops
        return ops.TIMER is None

    def _write_hyper(self):
        g = self.opt.param_groups[0]
        step = self.opt._step + 1
        from . import ops
        b1, b2 = g["betas"]
        ops.adamw_hyper(g["lr"], b1, b2, step, self.hyper_host)      # (formed in C exactly like the eager kernel's arguments)
        self.hyper.copy_(self.hyper_host, non_blocking=True)

    def _check_label_cap(self):
        """the count of the PREVIOUS replay (copied to pinned memory behind it): a batch with more labelled rows than the bound lost
        rows -- that step is wrong and says so, one step late, instead of costing every step a sync"""
        if self._count_ev is not None and self._count_ev.query():
            n, cap = int(self._count_host[0]), int(self.model.llm.label_rows_cap)
            self._count_ev = None
            if n > cap:
                raise RuntimeError(f"GraphedTrainStep: a batch carried {n} labelled rows, more than label_rows_cap = {cap}: the previous step dropped "
                                   "rows; construct GraphedTrainStep(label_rows_cap=...) with the collator's bound (B x longest answer)")

    def _body(self):
        from . import ops
        m = self.model
        outputs, acc = m(**self.static)
        loss = outputs.loss
        loss.backward()
        st, g = m.store, self.opt.param_groups[0]
        ops.adamw_step_dev(st.flat, self.opt._flat_grad(), self.opt.exp_avg, self.opt.exp_avg_sq, st.flat_bf16, self.hyper,
                           g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"])
        m.refresh_derived()
        m._always_refresh = False
        self.opt.zero_grad()
        self.salt.add_(self.SALT_STEP)
        return loss.detach(), acc

    # ---- the step ----
    def __call__(self, batch: dict):
        from . import ops
        self.calls += 1
        self._check_label_cap()
        m = self.model
        if self.calls <= self.warmup or not self._capturable(batch) or (self.key is not None and self._signature(batch) != self.key):
            if m.llm.label_rows_cap is None and self.cap_arg:
                m.llm.label_rows_cap = int(self.cap_arg)
            self.eager_steps += 1
            return train_step(m, batch, self.opt, self.sched, self.gsync)
        if self.graph is None:
            if m.llm.label_rows_cap is None:
                if self.cap_arg:
                    m.llm.label_rows_cap = int(self.cap_arg)
                elif batch.get("labels") is not None:      # (one sync, once: the capture batch's own count, rounded up)
                    n = int((batch["labels"][:, 1:] >= 0).sum())
                    m.llm.label_rows_cap = max(64, -(-n // 64) * 64)
            self.key = self._signature(batch)
            self.static = {k: v.clone() for k, v in batch.items()}
            self._write_hyper()
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            ops.set_dropout_salt(self.salt)
            try:
                with torch.cuda.graph(self.graph):
                    self.out = self._body()
            finally:
                ops.set_dropout_salt(None)       # (the pointer is a kernel argument of the captured launches; eager code keeps its plain seeds)
        else:
            for k, v in batch.items():
                self.static[k].copy_(v, non_blocking=True)
            self._write_hyper()
        self.graph.replay()
        self.replays += 1
        self.opt._step += 1
        if self.sched is not None:
            self.sched.step()
        if m.llm.last_label_count is not None:
            self._count_host.copy_(m.llm.last_label_count, non_blocking=True)
            self._count_ev = torch.cuda.Event()
            self._count_ev.record()
        return self.out


def lr_lambda(step: int, warmup_steps: int, total_steps: int) -> float:
    """linear warm-up then linear decay to 0 (src/slam_llm/pipeline/finetune.py:253-260)."""
    if step < warmup_steps:
        return min(step / warmup_steps, 1)
    return max(0.0, 1 - (step - warmup_steps) / (total_steps - warmup_steps))
