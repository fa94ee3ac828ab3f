"""Host-side mirror of the reference's encoder -> projector -> LLM(+LoRA) composite for the HIP path.

Mirrors `slam_llm.models.slam_model.slam_model` (src/slam_llm/models/slam_model.py:239-456): same constructor
surface (`.encoder`, `.encoder_projector`, `.llm`, `.tokenizer`), same `forward(**batch) -> (outputs, acc)` with
`outputs.loss` autograd-connected, `inference_mode=True -> (inputs_embeds, attention_mask)`, same state_dict
key names for the trainable tensors (`encoder_projector.linear1.weight`, peft's
`llm.base_model.model.model.layers.N.self_attn.q_proj.lora_A.default.weight`, ...).

Everything numerical is a C-ABI call into libslamhip.so (slam_llm_amd.ops); torch provides HBM buffers, the
HIP stream and the autograd hook (`loss.backward()` enters `_SlamStep.backward`, which runs the hand-written
backward kernels and deposits gradients into one flat fp32 buffer that `.grad` of every trainable parameter
views).  Data layout in HBM (sized for 288 GB):
  * frozen weights bf16 in BOTH orientations (W for forward, W^T for dX) -> every product is an NT GEMM;
  * LoRA is folded into the frozen GEMM by K-extension: y = [x | xA^T] . [W | (alpha/r)B]^T; the same
    extended W^T yields dx and d(xA^T) in one backward GEMM;
  * trainable parameters: one flat fp32 master buffer (+ flat fp32 grad, + bf16 compute copies), ordered in
    backward-production order (last LLM layer first, projector last) so data-parallel all-reduce can start
    on a prefix while the rest of the backward is still running.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops, trace
from .ops import ACT_GELU, ACT_NONE, ACT_RELU, round_up

import os as _os
# SwiGLU backward inside the down_proj dX GEMM epilogue (act = 3).  Correct and tested, but measured neutral-to-slower
# on the C3 step (A/B in one process: 444.2-446.4 ms fused vs 443.7-444.6 ms separate pass): with one 256x256 workgroup
# per CU the exp-heavy epilogue is not overlapped with MFMA work.  Off unless SLAM_FUSED_SWIGLU_BWD=1.
FUSE_SWIGLU_BWD = _os.environ.get("SLAM_FUSED_SWIGLU_BWD", "0") == "1"
# HuBERT / WavLM positional conv as one implicit-GEMM launch (slam_pos_conv_fwd); SLAM_POS_CONV_FUSED=0: 16 x (im2col + GEMM)
POS_CONV_FUSED = _os.environ.get("SLAM_POS_CONV_FUSED", "1") == "1"
# SwiGLU forward inside the gate|up product's epilogue (slam_gemm_swiglu_bf16_nt).  Correct (bit-identical to product -> swiglu_fwd) and
# tested, but measured NEUTRAL on the C3 step like its backward twin (three A/Bs on two boxes: -1.3, -0.5, +0.3 ms of ~394: the 5.5 ms
# elementwise pass it removes comes back as a 1.5x larger store tail in the 32 gate|up launches, 4-wave epilogue, one workgroup per
# CU) -- and it keeps a third, block-interleaved copy of every [gate ; up] weight (7.5 GB at Llama-3-8B).  Off unless
# SLAM_FUSED_SWIGLU_FWD=1.
FUSE_SWIGLU_FWD = _os.environ.get("SLAM_FUSED_SWIGLU_FWD", "0") == "1"
# Training steps run the lm_head product, the cross entropy and the lm_head dX product only over the rows that carry a label: rows whose
# (shifted) label is -100 -- the audio span, the prompt, the pad tokens: 83 % of the C3 batch -- enter neither the loss nor the accuracy
# (utils/metric.py compute_accuracy ignores them) nor any gradient (their dL/dlogits is exactly zero), and the training forward hands
# no logits out.  The label count reaches the host through a pinned buffer written at the START of the forward (an event wait that
# has long completed by the time the LLM reaches its head).  SLAM_LM_HEAD_LABEL_ROWS=0 computes every row like HF does.
LM_HEAD_LABEL_ROWS = _os.environ.get("SLAM_LM_HEAD_LABEL_ROWS", "1") == "1"
# ... and the LAST decoder layer needs its attention over every row (keys / values), but everything behind the attention -- o_proj, the
# residual, the MLP, the final RMSNorm -- only feeds the head: those run over the labelled rows too, forward and backward (the rows
# without a label get exactly the zero gradient they had).  Off when LoRA dropout acts on o / gate / up / down of that layer (the
# counter-based masks are indexed by row).  SLAM_LAST_LAYER_LABEL_ROWS=0 disables.
LAST_LAYER_LABEL_ROWS = _os.environ.get("SLAM_LAST_LAYER_LABEL_ROWS", "1") == "1"
# LoRA backward under lora_dropout: second hop + mask + accumulate as ONE pass over dx (slam_lora_hop_dropout) instead of product -> scratch
# [M, K] -> slam_dropout_bf16(accumulate); bit-identical; SLAM_FUSED_LORA_HOP=0 restores the two launches (A/B)
FUSE_LORA_HOP = _os.environ.get("SLAM_FUSED_LORA_HOP", "1") == "1"
# LoRA gradient products (dA / dB grams and their reduces) on a side stream under the next layer's GEMMs (round 6, VERDICT r5 next #5a):
# bit-identical; SLAM_LORA_SIDE_STREAM=0 keeps them inline on the compute stream (A/B)
LORA_SIDE_STREAM = _os.environ.get("SLAM_LORA_SIDE_STREAM", "0") == "1"
_SIDE_STREAMS = {}


def _side_stream(device):
    key = str(device)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


LORA_PAD = 64  # K-extension granule (GEMM K-tile)
# HuBERT / WavLM conv layers 1-6 (k 3 / 2, stride 2, 512 channels, no padding): the im2col matrix of a row-major [T, C] signal is a VIEW
# with overlapping rows (row t = the k * C contiguous elements from row 2 t on) -- handed to the GEMM as lda = stride * C, one launch
# per clip, instead of 2.9 GB of materialised columns per C4 step.  SLAM_CONV_WINDOW_VIEW=0: the im2col kernel + one launch per layer.
CONV_WINDOW_VIEW = _os.environ.get("SLAM_CONV_WINDOW_VIEW", "1") == "1"


# ======================================================================================== presets
PRESETS = {
    # encoders (openai-whisper dims)
    "whisper-tiny": dict(n_mels=80, enc_dim=384, enc_heads=6, enc_layers=4),
    "whisper-base": dict(n_mels=80, enc_dim=512, enc_heads=8, enc_layers=6),
    "whisper-small": dict(n_mels=80, enc_dim=768, enc_heads=12, enc_layers=12),
    "whisper-medium": dict(n_mels=80, enc_dim=1024, enc_heads=16, enc_layers=24),
    "whisper-large-v2": dict(n_mels=80, enc_dim=1280, enc_heads=20, enc_layers=32),
    "whisper-large-v3": dict(n_mels=128, enc_dim=1280, enc_heads=20, enc_layers=32),
    # LLMs
    "llama-3-8b": dict(llm_dim=4096, llm_layers=32, llm_heads=32, llm_kv_heads=8, llm_head_dim=128, llm_ffn=14336,
                       vocab=128256, rope_theta=500000.0, rms_eps=1e-5),
    "tinyllama-1.1b": dict(llm_dim=2048, llm_layers=22, llm_heads=32, llm_kv_heads=4, llm_head_dim=64, llm_ffn=5632,
                           vocab=32000, rope_theta=10000.0, rms_eps=1e-5),
    "vicuna-13b": dict(llm_dim=5120, llm_layers=40, llm_heads=40, llm_kv_heads=40, llm_head_dim=128, llm_ffn=13824,
                       vocab=32000, rope_theta=10000.0, rms_eps=1e-5),
    "vicuna-7b": dict(llm_dim=4096, llm_layers=32, llm_heads=32, llm_kv_heads=32, llm_head_dim=128, llm_ffn=11008,
                      vocab=32000, rope_theta=10000.0, rms_eps=1e-5),
}


def make_config(encoder: Optional[str] = None, llm: Optional[str] = None, **kw) -> dict:
    cfg = dict(n_mels=80, enc_dim=128, enc_heads=2, enc_layers=2, enc_ctx=1500, ds_rate=5, proj_hidden=2048,
               llm_dim=128, llm_layers=2, llm_heads=2, llm_kv_heads=1, llm_head_dim=64, llm_ffn=256, vocab=512,
               rope_theta=10000.0, rms_eps=1e-5, lora_r=8, lora_alpha=32, lora_targets=("q_proj", "v_proj"),
               lora_dropout=0.0)
    if encoder:
        cfg.update(PRESETS[encoder])
    if llm:
        cfg.update(PRESETS[llm])
    cfg.update(kw)
    return cfg


# ======================================================================================== trainable store
class TrainableStore:
    """One flat fp32 master / grad buffer (+ bf16 copy); parameters are views (64-element aligned)."""

    def __init__(self, device):
        self.device = device
        self.entries: List[Tuple[str, Tuple[int, ...], int]] = []
        self.size = 0
        self.flat = self.grad = self.flat_bf16 = None
        self.pure_bf16, self.grad_lp = False, None
        self.params: Dict[str, nn.Parameter] = {}

    def reserve(self, name: str, shape) -> None:
        assert self.flat is None
        n = int(math.prod(shape))
        self.entries.append((name, tuple(shape), self.size))
        self.size += round_up(n, 64)

    def allocate(self):
        self.flat = torch.zeros(self.size, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(self.size, dtype=torch.float32, device=self.device)
        self.flat_bf16 = torch.zeros(self.size, dtype=torch.bfloat16, device=self.device)
        self.offsets = {}
        for name, shape, off in self.entries:
            n = int(math.prod(shape))
            self.params[name] = nn.Parameter(self.flat[off:off + n].view(shape), requires_grad=True)
            self.offsets[name] = (off, n, shape)

    def grad_view(self, name):
        off, n, shape = self.offsets[name]
        return self.grad[off:off + n].view(shape)

    def bf16_view(self, name):
        off, n, shape = self.offsets[name]
        return self.flat_bf16[off:off + n].view(shape)

    def master_view(self, name):
        off, n, shape = self.offsets[name]
        return self.flat[off:off + n].view(shape)

    def refresh_bf16(self):
        if self.pure_bf16:      # the bf16 buffer IS the parameters: refresh their fp32 shadow (LayerNorm weights / biases are read in fp32)
            self.flat.copy_(self.flat_bf16)
        else:
            ops.cast_bf16(self.flat, self.flat_bf16)

    def to_pure_bf16(self):
        """`model.to(torch.bfloat16)` of the reference's pure_bf16 route (src/slam_llm/pipeline/finetune.py:154-155): every trainable
        parameter becomes a bf16 tensor -- a view of the flat bf16 buffer the kernels read anyway, so `torch.optim.AdamW(model.parameters())`
        or AnyPrecisionAdamW (finetune.py:237-251) update the operands of the next forward in place; the fp32 buffer turns into their
        up-cast shadow.  Gradients are handed out in bf16 (`.grad` dtype must match): the kernels still reduce in fp32, the flat fp32
        gradient buffer is rounded once per backward into `grad_lp`."""
        if self.pure_bf16:
            return
        with torch.no_grad():
            ops.cast_bf16(self.flat, self.flat_bf16)
            self.flat.copy_(self.flat_bf16)
            self.grad_lp = torch.zeros(self.size, dtype=torch.bfloat16, device=self.device)
            for name, (off, n, shape) in self.offsets.items():
                prm = self.params[name]
                prm.grad = None
                prm.data = self.flat_bf16[off:off + n].view(shape)
        self.pure_bf16 = True

    def grad_lp_view(self, name):
        off, n, shape = self.offsets[name]
        return self.grad_lp[off:off + n].view(shape)


# ======================================================================================== fused (LoRA) linear
class FusedLinear:
    """Frozen y = x W^T for a group of projections sharing the input (q|k|v, gate|up, ...), with optional
    peft-style LoRA adapters on any member folded in by K-extension (see module docstring)."""

    def __init__(self, K: int, parts: List[Tuple[str, int]], device):
        self.K = K
        self.parts = parts
        self.N = sum(r for _, r in parts)
        self.row0 = {}
        o = 0
        for n, r in parts:
            self.row0[n] = o
            o += r
        self.device = device
        self.adapters = []  # dicts: part, row0, rows, r, j0, scale, A(name), B(name)
        self.Rp = 0
        self.Wext = self.WextT = None
        self._base = {}

    def set_base(self, part: str, w: torch.Tensor):
        self._base[part] = w

    def add_lora(self, part: str, r: int, alpha: float, a_name: str, b_name: str):
        rows = dict(self.parts)[part]
        j0 = sum(a["r"] for a in self.adapters)
        self.adapters.append(dict(part=part, row0=self.row0[part], rows=rows, r=r, j0=j0, scale=alpha / r,
                                  A=a_name, B=b_name))
        self.Rp = round_up(j0 + r, LORA_PAD)

    @property
    def sum_r(self):
        return sum(a["r"] for a in self.adapters)

    def finalize(self):
        """materialise [W | 0] and its transpose from the parts' base weights (bf16, device)."""
        Kx = self.K + self.Rp
        assert self.N % 64 == 0 and self.K % 64 == 0, f"fused linear dims must be multiples of 64 (N={self.N}, K={self.K})"
        self.Wext = torch.zeros((self.N, Kx), dtype=torch.bfloat16, device=self.device)
        for n, rows in self.parts:
            w = self._base.pop(n)
            self.Wext[self.row0[n]: self.row0[n] + rows, : self.K].copy_(w)
        self.WextT = torch.zeros((Kx, self.N), dtype=torch.bfloat16, device=self.device)
        ops.transpose(self.Wext[:, : self.K], Rp=self.N, out=self.WextT[: self.K])
        # a frozen [gate ; up] group without adapters also keeps its rows block-interleaved for the fused SwiGLU-forward product
        # (288 GB: a third copy of these 235 MB per Llama-3-8B layer is free; the decode / eval paths keep using Wext)
        self.Wil = None
        if FUSE_SWIGLU_FWD and [n for n, _ in self.parts] == ["gate_proj", "up_proj"] and not self.adapters and (self.N // 2) % 64 == 0:
            self.Wil = ops.interleave_gate_up(self.Wext)

    def refresh(self, store: TrainableStore):
        """re-pack the adapters from the fp32 masters (after an optimizer step)."""
        if not self.adapters:
            return
        K = self.K
        for a in self.adapters:
            ops.lora_pack_b(store.master_view(a["B"]), a["scale"],
                            self.Wext[a["row0"]: a["row0"] + a["rows"], K + a["j0"]: K + a["j0"] + a["r"]],
                            self.WextT[K + a["j0"]: K + a["j0"] + a["r"], a["row0"]: a["row0"] + a["rows"]])
        # A's bf16 copies live in the store's flat bf16 buffer (contiguous block of all adapters of this group);
        # their transpose [K, Rp] serves the second hop of the backward: dx += d(xA^T) . A
        self.AcatT = ops.transpose_into(getattr(self, "AcatT", None), self.a_cat(store), self.Rp)     # in place: see ops.transpose_into

    def a_cat(self, store: TrainableStore) -> torch.Tensor:
        first = self.adapters[0]
        off, _, _ = store.offsets[first["A"]]
        return store.flat_bf16[off: off + self.sum_r * self.K].view(self.sum_r, self.K)

    def a_cat_grad(self, store: TrainableStore) -> torch.Tensor:
        first = self.adapters[0]
        off, _, _ = store.offsets[first["A"]]
        return store.grad[off: off + self.sum_r * self.K].view(self.sum_r, self.K)

    @property
    def hop_fills_pad(self) -> bool:
        """forward() runs the first hop through slam_lora_a_fwd, which also zeroes the padding columns of the K-extension"""
        return bool(self.adapters) and self.sum_r <= 64 and self.sum_r % 4 == 0

    def new_input(self, M: int, for_forward: bool = False) -> torch.Tensor:
        """[M, K + Rp] activation buffer; the caller fills [:, :K].  for_forward: the buffer goes straight into forward(), whose first-hop
        kernel defines every extension column (results + zero padding) -- no fill launch here (round 5: 32 strided fills per C3 step)."""
        buf = torch.empty((M, self.K + self.Rp), dtype=torch.bfloat16, device=self.device)
        if self.Rp and self.Rp > self.sum_r and not (for_forward and self.hop_fills_pad):
            buf[:, self.K + self.sum_r:].zero_()  # padding columns of the K-extension
        return buf

    def forward(self, x_ext: torch.Tensor, store: Optional[TrainableStore], out=None, residual=None, bias=None,
                act=ACT_NONE, drop=None):
        """drop = (p, seed, offset) applies peft's lora_dropout to the adapters' input (one mask per fused group:
        peft samples one mask per adapted Linear -- sharing it between e.g. q_proj and v_proj of a layer is a stated
        deviation); None = no dropout (eval mode / p = 0)."""
        if self.adapters:
            sr = self.sum_r
            if self.hop_fills_pad:   # one pass over x, dropout mask applied in registers; the padding columns zeroed by the same launch
                ops.lora_a_fwd(x_ext[:, : self.K], self.a_cat(store), x_ext[:, self.K: self.K + sr], drop, pad_to=self.Rp)
            else:
                xin = x_ext[:, : self.K]
                if drop is not None:
                    xin = ops.dropout(xin, *drop)
                ops.gemm_nt(xin, self.a_cat(store), out=x_ext[:, self.K: self.K + sr])
        return ops.gemm_nt(x_ext, self.Wext, out=out, residual=residual, bias=bias, act=act,
                           k_alg=self.K + self.sum_r)

    def forward_swiglu(self, x_ext: torch.Tensor, store: Optional[TrainableStore], out: torch.Tensor):
        """decode (<= 64 rows): this group is [gate | up]; out = silu(gate) * up from one weight-streaming launch."""
        if self.adapters:
            ops.gemm_nt(x_ext[:, : self.K], self.a_cat(store), out=x_ext[:, self.K: self.K + self.sum_r])
        return ops.gemm_skinny(x_ext, self.Wext, out, swiglu=True)

    def backward(self, dy: torch.Tensor, x_ext: Optional[torch.Tensor], store: Optional[TrainableStore],
                 accumulate: bool, out=None, drop=None, defer=None) -> torch.Tensor:
        """returns dx_ext [M, K+Rp] (columns [:K] are dL/dx); deposits adapter gradients into the store.
        defer(fn, keep): run the adapter-gradient products (off the dX critical path) through the caller's side queue instead of inline."""
        if self.Rp and self.K % 256 == 0:
            # two products instead of one [M, K + Rp] output: the Rp (64) extension columns would open a 17th column of 256-wide
            # tiles that is 25 % full -- at the Llama qkv shape 799 tiles = 3.12 rounds over 256 CUs, paid as 4 (measured 1.1 PF
            # vs 1.3 for its neighbours).  K columns = whole tiles, the extension = one narrow (N <= 64) product.
            dx_ext = out if out is not None else torch.empty((dy.shape[0], self.K + self.Rp), dtype=torch.bfloat16, device=dy.device)
            ops.gemm_nt(dy, self.WextT[: self.K], out=dx_ext[:, : self.K])
            ops.gemm_nt(dy, self.WextT[self.K:], out=dx_ext[:, self.K:])    # (through the first-hop kernel instead: same step time, A/B'd)
        else:
            dx_ext = ops.gemm_nt(dy, self.WextT, out=out)
        if self.adapters:
            # x_ext = [x | u] with u = dropout(x) A^T: the columns [K:] of dx_ext are dL/du, and dL/dx gets the second hop
            xin = x_ext[:, : self.K]
            if drop is None:
                ops.gemm_nt(dx_ext[:, self.K:], self.AcatT, out=dx_ext[:, : self.K], accumulate=True)
            elif FUSE_LORA_HOP and self.Rp in (32, 64):
                # dL/dx += mask . (du . A) / (1 - p): the rank-Rp product, the mask (recomputed) and the accumulate in one pass over dx
                ops.lora_hop_dropout(dx_ext[:, self.K:], self.AcatT, dx_ext[:, : self.K], drop)
            else:
                hop = ops.gemm_nt(dx_ext[:, self.K:], self.AcatT)           # dL/d(dropout(x))
                ops.dropout(hop, *drop, out=dx_ext[:, : self.K], accumulate=True)  # same mask, recomputed
            K, sr = self.K, self.sum_r

            def adapter_grads():
                # dA[r, K] = du^T x ;  dB[rows, r] = (alpha/r) * dy[:, rows]^T u   (HBM-bound tall-skinny products).
                # All adapters of the group share x and their A's (and A gradients) are contiguous: one pass over x.
                # (dropout(x) is never materialised: the gram kernel recomputes the mask)
                merged = sr % 8 == 0 and sr <= 64
                if merged:
                    ops.skinny_gram(dx_ext[:, K: K + sr], xin, self.a_cat_grad(store), K, 1, accumulate=accumulate, drop=drop)
                for a in self.adapters:
                    r = a["r"]
                    if not merged:
                        ops.skinny_gram(dx_ext[:, K + a["j0"]: K + a["j0"] + r], xin, store.grad_view(a["A"]), K, 1,
                                        accumulate=accumulate, drop=drop)
                    u = x_ext[:, K + a["j0"]: K + a["j0"] + r]            # xA^T        [M, r]
                    ops.skinny_gram(u, dy[:, a["row0"]: a["row0"] + a["rows"]], store.grad_view(a["B"]), 1, r,
                                    alpha=a["scale"], accumulate=accumulate)
            if defer is not None:
                # NB the second hop above only ADDS into dx_ext[:, :K]; the grams read dx_ext[:, K:] (du), x and dy, none of which the main
                # stream writes again -- the side queue only has to keep them alive until it is joined
                defer(adapter_grads, (dx_ext, x_ext, dy))
            else:
                adapter_grads()
        return dx_ext


# ======================================================================================== whisper encoder
class HipWhisperEncoder(nn.Module):
    """Whisper audio encoder (src/slam_llm/models/encoder.py:13-30).  bf16 weights/activations, LayerNorm
    statistics in fp32 (deviation from the fp32 reference: stated in DESIGN.md).

    Frozen (store=None, the recipes' `freeze_encoder=true`): inference-only, weights held as bf16 compute copies.
    Trainable (store given, `train_config.freeze_encoder=false`, models/slam_model.py:110-113): every parameter of the
    reference module lives in the flat fp32 store under its reference name (`encoder.blocks.N.attn.query.weight`, ...; the
    sinusoidal `positional_embedding` is a buffer there and stays fixed), forward_train() keeps what the hand-written
    backward_hip() needs, and the bf16 compute copies (`self.w`) are rebuilt from the store by refresh() after each step."""

    QFOLD = ops.qscale(64 ** -0.5)   # what the FROZEN encoder's query projection is multiplied by at load time (head_dim 64)

    def __init__(self, cfg: dict, device, store: Optional["TrainableStore"] = None, prefix="encoder."):
        super().__init__()
        self.cfg = cfg
        self.device_ = device
        self.w = {}
        self.store, self.prefix = store, prefix
        if store is not None:
            d, nm = cfg["enc_dim"], cfg["n_mels"]
            assert d % 64 == 0 and d // cfg["enc_heads"] == 64, "whisper head_dim must be 64"
            r, p = store.reserve, prefix
            # reserved in the order the backward produces them (GradSync prefixes): ln_post, blocks last -> first, convs
            r(p + "ln_post.weight", (d,)); r(p + "ln_post.bias", (d,))
            for i in reversed(range(cfg["enc_layers"])):
                b = f"{p}blocks.{i}."
                r(b + "mlp.2.weight", (d, 4 * d)); r(b + "mlp.2.bias", (d,))
                r(b + "mlp.0.weight", (4 * d, d)); r(b + "mlp.0.bias", (4 * d,))
                r(b + "mlp_ln.weight", (d,)); r(b + "mlp_ln.bias", (d,))
                r(b + "attn.out.weight", (d, d)); r(b + "attn.out.bias", (d,))
                for n in ("query", "key", "value"):   # back to back: the bf16 copies form the fused [3d, d] operand
                    r(b + f"attn.{n}.weight", (d, d))
                r(b + "attn.query.bias", (d,)); r(b + "attn.value.bias", (d,))   # (the key projection has no bias)
                r(b + "attn_ln.weight", (d,)); r(b + "attn_ln.bias", (d,))
            r(p + "conv2.weight", (d, d, 3)); r(p + "conv2.bias", (d,))
            r(p + "conv1.weight", (d, nm, 3)); r(p + "conv1.bias", (d,))

    @property
    def trainable(self) -> bool:
        return self.store is not None

    @torch.no_grad()
    def refold_query_bias(self):
        """frozen encoder: rebuild the scaled query bias from the unscaled one (after `model.to(torch.bfloat16)` rounded the frozen fp32
        tensors: the reference's cast rounds the bias ITSELF, the scale is then applied to the rounded value in fp32)"""
        if self.trainable:
            return
        d = self.cfg["enc_dim"]
        for i in range(self.cfg["enc_layers"]):
            self.w[f"{i}.qkv_b"][:d] = self.w[f"{i}.q_b_raw"] * self.QFOLD

    def bind(self):
        for name, prm in self.store.params.items():
            if name.startswith(self.prefix):
                _attach(self, name[len(self.prefix):], prm)

    def load(self, W: Dict[str, torch.Tensor], prefix="encoder."):
        cfg, dev = self.cfg, self.device_
        if self.trainable:
            with torch.no_grad():
                for name, prm in self.store.params.items():
                    if name.startswith(self.prefix):
                        prm.copy_(W[prefix + name[len(self.prefix):]].to(dev))
            self.w["pos"] = W[prefix + "positional_embedding"].to(device=dev, dtype=torch.bfloat16).contiguous()
            self.kp1 = round_up(3 * cfg["n_mels"], 64)
            return self
        d, nm = cfg["enc_dim"], cfg["n_mels"]
        assert d % 64 == 0 and d // cfg["enc_heads"] == 64, "whisper head_dim must be 64"
        bf = lambda t: t.to(device=dev, dtype=torch.bfloat16).contiguous()  # noqa: E731
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        w = self.w
        self.kp1 = round_up(3 * nm, 64)
        c1 = torch.zeros((d, self.kp1), dtype=torch.bfloat16, device=dev)
        c1[:, : 3 * nm] = bf(W[prefix + "conv1.weight"].permute(0, 2, 1).reshape(d, 3 * nm))
        w["conv1"], w["conv1_b"] = c1, f32(W[prefix + "conv1.bias"])
        w["conv2"] = bf(W[prefix + "conv2.weight"].permute(0, 2, 1).reshape(d, 3 * d))
        w["conv2_b"] = f32(W[prefix + "conv2.bias"])
        w["pos"] = bf(W[prefix + "positional_embedding"])
        for i in range(cfg["enc_layers"]):
            p = f"{prefix}blocks.{i}."
            # the frozen query projection carries the softmax scale and the exponent's base change (QFOLD = d_head^-1/2 * log2 e, folded in
            # fp32 BEFORE the one rounding to bf16): attn_fwd(..., q_prescaled=True) then takes the scores as they come out of the product
            # (an un-frozen encoder keeps the reference's own weights: its eval forward passes q_prescaled=False)
            w[f"{i}.qkv"] = bf(torch.cat([W[p + "attn.query.weight"].float() * self.QFOLD, W[p + "attn.key.weight"], W[p + "attn.value.weight"]], 0))
            w[f"{i}.qkv_b"] = f32(torch.cat([W[p + "attn.query.bias"].float() * self.QFOLD, torch.zeros(d), W[p + "attn.value.bias"]], 0))
            w[f"{i}.q_b_raw"] = f32(W[p + "attn.query.bias"])     # (the unscaled bias: refold_query_bias)
            w[f"{i}.out"], w[f"{i}.out_b"] = bf(W[p + "attn.out.weight"]), f32(W[p + "attn.out.bias"])
            w[f"{i}.ln1_w"], w[f"{i}.ln1_b"] = f32(W[p + "attn_ln.weight"]), f32(W[p + "attn_ln.bias"])
            w[f"{i}.fc1"], w[f"{i}.fc1_b"] = bf(W[p + "mlp.0.weight"]), f32(W[p + "mlp.0.bias"])
            w[f"{i}.fc2"], w[f"{i}.fc2_b"] = bf(W[p + "mlp.2.weight"]), f32(W[p + "mlp.2.bias"])
            w[f"{i}.ln2_w"], w[f"{i}.ln2_b"] = f32(W[p + "mlp_ln.weight"]), f32(W[p + "mlp_ln.bias"])
        w["lnp_w"], w["lnp_b"] = f32(W[prefix + "ln_post.weight"]), f32(W[prefix + "ln_post.bias"])
        return self

    def init_random(self, seed: int = 42):
        """seeded random weights generated directly in HBM at the true dimensions (benchmarks: no checkpoints offline)."""
        cfg, dev = self.cfg, self.device_
        d, nm = cfg["enc_dim"], cfg["n_mels"]
        g = torch.Generator(device=dev).manual_seed(seed)
        rn = lambda *s, std=0.02: (torch.randn(*s, generator=g, device=dev) * std)  # noqa: E731
        w = self.w
        self.kp1 = round_up(3 * nm, 64)
        if self.trainable:
            from .host_tables import sinusoids
            w["pos"] = sinusoids(cfg["enc_ctx"], d).to(device=dev, dtype=torch.bfloat16)
            with torch.no_grad():
                for name, prm in self.store.params.items():
                    if not name.startswith(self.prefix):
                        continue
                    if name.endswith("_ln.weight") or name.endswith("ln_post.weight"):
                        prm.copy_(1 + rn(*prm.shape, std=0.1))
                    elif name.endswith(".bias"):
                        prm.copy_(rn(*prm.shape, std=0.1 if "ln" in name.rsplit(".", 2)[-2] else 0.02))
                    else:
                        fan_in = int(math.prod(prm.shape[1:]))
                        prm.copy_(rn(*prm.shape, std=fan_in ** -0.5))
            return self
        c1 = torch.zeros((d, self.kp1), dtype=torch.bfloat16, device=dev)
        c1[:, : 3 * nm] = rn(d, 3 * nm, std=(3 * nm) ** -0.5).to(torch.bfloat16)
        w["conv1"], w["conv1_b"] = c1, rn(d)
        w["conv2"], w["conv2_b"] = rn(d, 3 * d, std=(3 * d) ** -0.5).to(torch.bfloat16), rn(d)
        from .host_tables import sinusoids
        w["pos"] = sinusoids(cfg["enc_ctx"], d).to(device=dev, dtype=torch.bfloat16)
        for i in range(cfg["enc_layers"]):
            wq = rn(3 * d, d, std=d ** -0.5)
            wq[:d] *= self.QFOLD            # (see load(): the frozen query projection carries scale * log2 e)
            w[f"{i}.qkv"] = wq.to(torch.bfloat16)
            del wq
            qb = rn(3 * d)
            w[f"{i}.q_b_raw"] = qb[:d].clone()
            qb[:d] *= self.QFOLD
            qb[d: 2 * d] = 0
            w[f"{i}.qkv_b"] = qb
            w[f"{i}.out"], w[f"{i}.out_b"] = rn(d, d, std=d ** -0.5).to(torch.bfloat16), rn(d)
            w[f"{i}.ln1_w"], w[f"{i}.ln1_b"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
            w[f"{i}.fc1"], w[f"{i}.fc1_b"] = rn(4 * d, d, std=d ** -0.5).to(torch.bfloat16), rn(4 * d)
            w[f"{i}.fc2"], w[f"{i}.fc2_b"] = rn(d, 4 * d, std=(4 * d) ** -0.5).to(torch.bfloat16), rn(d)
            w[f"{i}.ln2_w"], w[f"{i}.ln2_b"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
        w["lnp_w"], w["lnp_b"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
        return self

    @torch.no_grad()
    def extract_variable_length_features(self, x: torch.Tensor) -> torch.Tensor:
        """x: [B, n_mels, T] (the reference passes audio_mel.permute(0, 2, 1)) -> [B, ceil(T/2), d] bf16."""
        mel = x.permute(0, 2, 1).contiguous()  # [B, T, n_mels]; a no-op copy for the reference's permuted view
        return self.forward_btc(mel)

    @torch.no_grad()
    def forward_btc(self, mel: torch.Tensor) -> torch.Tensor:
        cfg, w = self.cfg, self.w
        B, T, nm = mel.shape
        d, H = cfg["enc_dim"], cfg["enc_heads"]
        cols = ops.conv1d_k3_im2col(mel, 1, self.kp1)
        h1 = ops.gemm_nt(cols, w["conv1"], bias=w["conv1_b"], act=ACT_GELU)
        del cols
        cols2 = ops.conv1d_k3_im2col(h1.view(B, T, d), 2, 3 * d)
        T2 = (T + 1) // 2
        assert T2 <= w["pos"].shape[0], "audio longer than the encoder's positional table"
        x = ops.gemm_nt(cols2, w["conv2"], bias=w["conv2_b"], act=ACT_GELU, residual=w["pos"], res_row_mod=T2)
        del cols2, h1
        M = B * T2
        scale = 64 ** -0.5
        hbuf = torch.empty((M, d), dtype=torch.bfloat16, device=mel.device)
        qkv = torch.empty((M, 3 * d), dtype=torch.bfloat16, device=mel.device)
        obuf = torch.empty((M, d), dtype=torch.bfloat16, device=mel.device)
        fbuf = torch.empty((M, 4 * d), dtype=torch.bfloat16, device=mel.device)
        for i in range(cfg["enc_layers"]):
            ops.layernorm(x, w[f"{i}.ln1_w"], w[f"{i}.ln1_b"], out=hbuf)
            ops.gemm_nt(hbuf, w[f"{i}.qkv"], out=qkv, bias=w[f"{i}.qkv_b"])
            ops.attn_fwd(qkv[:, :d], qkv[:, d: 2 * d], qkv[:, 2 * d:], B, T2, H, H, 64, False, scale, want_lse=False, out=obuf, q_prescaled=not self.trainable)
            ops.gemm_nt(obuf, w[f"{i}.out"], out=x, bias=w[f"{i}.out_b"], residual=x)
            ops.layernorm(x, w[f"{i}.ln2_w"], w[f"{i}.ln2_b"], out=hbuf)
            ops.gemm_nt(hbuf, w[f"{i}.fc1"], out=fbuf, bias=w[f"{i}.fc1_b"], act=ACT_GELU)
            ops.gemm_nt(fbuf, w[f"{i}.fc2"], out=x, bias=w[f"{i}.fc2_b"], residual=x)
        out = ops.layernorm(x, w["lnp_w"], w["lnp_b"])
        return out.view(B, T2, d)

    def forward(self, x):
        return self.extract_variable_length_features(x)

    # ---- trainable mode ----------------------------------------------------------------------------------
    def refresh(self):
        """rebuild the bf16 compute copies (and the transposes the backward multiplies by) from the store; called with the
        model's refresh after every optimizer step"""
        if not self.trainable:
            return
        st, p, cfg, w = self.store, self.prefix, self.cfg, self.w
        d, nm = cfg["enc_dim"], cfg["n_mels"]
        self.kp1 = round_up(3 * nm, 64)
        c1 = torch.zeros((d, self.kp1), dtype=torch.bfloat16, device=self.device_)
        c1[:, : 3 * nm] = st.bf16_view(p + "conv1.weight").permute(0, 2, 1).reshape(d, 3 * nm)   # tap-major columns (im2col order)
        w["conv1"], w["conv1_b"] = c1, st.master_view(p + "conv1.bias")
        w["conv2"] = st.bf16_view(p + "conv2.weight").permute(0, 2, 1).reshape(d, 3 * d).contiguous()
        w["conv2_b"] = st.master_view(p + "conv2.bias")
        w["conv2T"] = ops.transpose(w["conv2"], Rp=d)                # [3d, d]: dcols = dz . W
        for i in range(cfg["enc_layers"]):
            b = f"{p}blocks.{i}."
            off = st.offsets[b + "attn.query.weight"][0]
            w[f"{i}.qkv"] = st.flat_bf16[off: off + 3 * d * d].view(3 * d, d)
            qb = torch.zeros((3 * d,), dtype=torch.float32, device=self.device_)
            qb[:d] = st.master_view(b + "attn.query.bias")
            qb[2 * d:] = st.master_view(b + "attn.value.bias")
            w[f"{i}.qkv_b"] = qb
            w[f"{i}.out"], w[f"{i}.out_b"] = st.bf16_view(b + "attn.out.weight"), st.master_view(b + "attn.out.bias")
            w[f"{i}.ln1_w"], w[f"{i}.ln1_b"] = st.master_view(b + "attn_ln.weight"), st.master_view(b + "attn_ln.bias")
            w[f"{i}.fc1"], w[f"{i}.fc1_b"] = st.bf16_view(b + "mlp.0.weight"), st.master_view(b + "mlp.0.bias")
            w[f"{i}.fc2"], w[f"{i}.fc2_b"] = st.bf16_view(b + "mlp.2.weight"), st.master_view(b + "mlp.2.bias")
            w[f"{i}.ln2_w"], w[f"{i}.ln2_b"] = st.master_view(b + "mlp_ln.weight"), st.master_view(b + "mlp_ln.bias")
            for nme in ("qkv", "out", "fc1", "fc2"):
                w[f"{i}.{nme}T"] = ops.transpose(w[f"{i}.{nme}"], Rp=w[f"{i}.{nme}"].shape[0])
        w["lnp_w"], w["lnp_b"] = st.master_view(p + "ln_post.weight"), st.master_view(p + "ln_post.bias")

    def _attn_fwd_geom(self, qkv: torch.Tensor, geom):
        """bidirectional self-attention of the blocks for the two row layouts of the trainable encoder: ("batch", B, T2) = B equal-length
        clips in one launch; ("clips", [(first row, rows), ...]) = the packed ragged layout, one launch per clip on row slices of the
        packed buffers (each clip attends to its own frames only, exactly as if alone in the batch).  Returns (a [M, d], lse handle)."""
        d, H = self.cfg["enc_dim"], self.cfg["enc_heads"]
        scale = 64 ** -0.5
        if geom[0] == "batch":
            _, B, T2 = geom
            return ops.attn_fwd(qkv[:, :d], qkv[:, d: 2 * d], qkv[:, 2 * d:], B, T2, H, H, 64, False, scale, want_lse=True)
        a = torch.empty((qkv.shape[0], d), dtype=torch.bfloat16, device=qkv.device)
        lses = []
        for r0, n in geom[1]:
            q = qkv[r0: r0 + n]
            _, lse = ops.attn_fwd(q[:, :d], q[:, d: 2 * d], q[:, 2 * d:], 1, n, H, H, 64, False, scale, want_lse=True, out=a[r0: r0 + n])
            lses.append(lse)
        return a, lses

    def _attn_bwd_geom(self, qkv, a, da, lse, dqkv, geom):
        d, H = self.cfg["enc_dim"], self.cfg["enc_heads"]
        scale = 64 ** -0.5
        if geom[0] == "batch":
            _, B, T2 = geom
            ops.attn_bwd(qkv[:, :d], qkv[:, d: 2 * d], qkv[:, 2 * d:], a, da, lse, dqkv[:, :d], dqkv[:, d: 2 * d], dqkv[:, 2 * d:],
                         B, T2, H, H, 64, False, scale)
            return
        for (r0, n), lse_c in zip(geom[1], lse):
            q, dq = qkv[r0: r0 + n], dqkv[r0: r0 + n]
            ops.attn_bwd(q[:, :d], q[:, d: 2 * d], q[:, 2 * d:], a[r0: r0 + n], da[r0: r0 + n], lse_c, dq[:, :d], dq[:, d: 2 * d], dq[:, 2 * d:],
                         1, n, H, H, 64, False, scale)

    def _blocks_train(self, x: torch.Tensor, S: dict, geom):
        """the transformer blocks + ln_post on rows x [M, d], keeping what _blocks_backward needs in S"""
        cfg, w = self.cfg, self.w
        S["blocks"], S["geom"] = [], geom
        for i in range(cfg["enc_layers"]):
            h, m1, r1 = ops.layernorm(x, w[f"{i}.ln1_w"], w[f"{i}.ln1_b"], stats=True)
            qkv = ops.gemm_nt(h, w[f"{i}.qkv"], bias=w[f"{i}.qkv_b"])
            a, lse = self._attn_fwd_geom(qkv, geom)
            x1 = ops.gemm_nt(a, w[f"{i}.out"], bias=w[f"{i}.out_b"], residual=x)
            h2, m2, r2 = ops.layernorm(x1, w[f"{i}.ln2_w"], w[f"{i}.ln2_b"], stats=True)
            z = ops.gemm_nt(h2, w[f"{i}.fc1"], bias=w[f"{i}.fc1_b"])
            x2 = ops.gemm_nt(ops.gelu_fwd(z), w[f"{i}.fc2"], bias=w[f"{i}.fc2_b"], residual=x1)
            S["blocks"].append(dict(x=x, m1=m1, r1=r1, h=h, qkv=qkv, a=a, lse=lse, x1=x1, m2=m2, r2=r2, h2=h2, z=z))
            x = x2
        out, mo, ro = ops.layernorm(x, w["lnp_w"], w["lnp_b"], stats=True)
        S.update(x_last=x, mo=mo, ro=ro)
        return out

    def forward_train(self, mel: torch.Tensor, stash: dict) -> torch.Tensor:
        """forward_btc that keeps the activations backward_hip() needs in stash["encoder"] (GELU pre-activations instead of
        fused-epilogue outputs; GELU outputs, im2col matrices and head transposes are recomputed in the backward).
        Stash size: 11 [M, d]-sized bf16 tensors per block (M = B * T2): 42 GB for 31 x 30 s through Whisper-large-v3."""
        cfg, w = self.cfg, self.w
        B, T, nm = mel.shape
        d = cfg["enc_dim"]
        T2 = (T + 1) // 2
        assert T2 <= w["pos"].shape[0], "audio longer than the encoder's positional table"
        M = B * T2
        z1 = ops.gemm_nt(ops.conv1d_k3_im2col(mel, 1, self.kp1), w["conv1"], bias=w["conv1_b"])
        h1 = ops.gelu_fwd(z1)
        z2 = ops.gemm_nt(ops.conv1d_k3_im2col(h1.view(B, T, d), 2, 3 * d), w["conv2"], bias=w["conv2_b"])
        del h1
        x = (ops.gelu_fwd(z2).view(B, T2, d) + w["pos"][:T2]).view(M, d)
        S = {"mel": mel, "z1": z1, "z2": z2, "B": B, "T": T, "T2": T2}
        out = self._blocks_train(x, S, ("batch", B, T2))
        stash["encoder"] = S
        return out.view(B, T2, d)

    def forward_packed_train(self, mel: torch.Tensor, n_frames: List[int], stash: dict):
        """forward_packed with the encoder UN-frozen (round 5: `train_config.freeze_encoder=false` + `++model_config.varlen_encoder=true`,
        which SlamHipModel used to refuse): same arithmetic as forward_packed -- every clip encoded as if alone in the batch -- keeping
        what backward_hip needs.  Attention runs one launch per clip on row slices of the packed buffers (the backward kernels take
        packed segments only under a causal mask); everything else runs on the packed rows.  Returns (packed output [sum T2_b, d], T2)."""
        cfg, w = self.cfg, self.w
        B, T, nm = mel.shape
        d, dev = cfg["enc_dim"], mel.device
        assert len(n_frames) == B and max(n_frames) <= T
        T2 = [(n + 1) // 2 for n in n_frames]
        T2max = (T + 1) // 2
        assert max(T2) <= w["pos"].shape[0], "audio longer than the encoder's positional table"
        nf = torch.tensor(n_frames, dtype=torch.int32).to(dev, non_blocking=True)
        z1 = ops.gemm_nt(ops.conv1d_k3_im2col(mel, 1, self.kp1, n_valid=nf), w["conv1"], bias=w["conv1_b"])
        h1 = ops.gelu_fwd(z1)
        z2 = ops.gemm_nt(ops.conv1d_k3_im2col(h1.view(B, T, d), 2, 3 * d, n_valid=nf), w["conv2"], bias=w["conv2_b"])   # frames past a clip's own read as zero
        del h1
        xpad = (ops.gelu_fwd(z2).view(B, T2max, d) + w["pos"][:T2max]).view(B * T2max, d)     # the positional embedding restarts at every clip
        idx, clips, inv2, valid1 = [], [], torch.full((B * T2max,), -1, dtype=torch.int32), torch.full((B * T,), -1, dtype=torch.int32)
        acc = 0
        for b_, (t2, n1) in enumerate(zip(T2, n_frames)):
            idx.append(torch.arange(b_ * T2max, b_ * T2max + t2, dtype=torch.int32))
            inv2[b_ * T2max: b_ * T2max + t2] = torch.arange(acc, acc + t2, dtype=torch.int32)   # padded conv2 row -> packed row
            valid1[b_ * T: b_ * T + n1] = torch.arange(b_ * T, b_ * T + n1, dtype=torch.int32)    # conv1 frames conv2 really read
            clips.append((acc, t2))
            acc += t2
        meta = torch.cat(idx).to(dev, non_blocking=True)
        x = ops.gather_rows(xpad, meta)
        del xpad
        S = {"mel": mel, "z1": z1, "z2": z2, "B": B, "T": T, "T2": T2max, "nf": nf, "inv2": inv2.to(dev, non_blocking=True),
             "valid1": valid1.to(dev, non_blocking=True)}
        out = self._blocks_train(x, S, ("clips", clips))
        stash["encoder"] = S
        return out, T2

    def _lin_grads(self, dy: torch.Tensor, x: torch.Tensor, w_name: str, acc: bool, N: Optional[int] = None,
                   K: Optional[int] = None, bias: Tuple = ()):
        """dW (+)= dy^T x into the flat gradient buffer at parameter `w_name` (N x K rows may span the fused q|k|v block);
        bias = ((param name, first column of dy, width), ...) column sums"""
        st = self.store
        M = dy.shape[0]
        Mp = round_up(M, 64)
        off, _, shape = st.offsets[w_name]
        N, K = N or shape[0], K or shape[1]
        ops.gemm_nt(ops.transpose(dy, Rp=Mp), ops.transpose(x, Rp=Mp), out=st.grad[off: off + N * K].view(N, K), accumulate=acc)
        for name, c0, n in bias:
            ops.colsum(dy[:, c0: c0 + n], st.grad_view(name), accumulate=acc)

    def backward_hip(self, dout: torch.Tensor, stash: dict, acc: bool):
        """dout [M, d] bf16 = dL/d(encoder output) on the rows the forward returned (B * T2 padded rows, or the packed rows of
        forward_packed_train); deposits every encoder gradient into the flat grad buffer (hand-written adjoint of
        extract_variable_length_features, models/encoder.py:13-30)"""
        cfg, w, st, p = self.cfg, self.w, self.store, self.prefix
        S = stash.pop("encoder")
        B, T, T2 = S["B"], S["T"], S["T2"]
        d, nm = cfg["enc_dim"], cfg["n_mels"]
        M = B * T2
        nf, geom = S.get("nf"), S["geom"]
        gv = st.grad_view
        dx = ops.layernorm_bwd(S["x_last"], S["mo"], S["ro"], w["lnp_w"], dout, dgamma=gv(p + "ln_post.weight"),
                               dbeta=gv(p + "ln_post.bias"), accumulate=acc)
        for i in reversed(range(cfg["enc_layers"])):
            R = S["blocks"][i]
            b = f"{p}blocks.{i}."
            # MLP: x2 = x1 + fc2(gelu(fc1(LN2(x1))))
            f = ops.gelu_fwd(R["z"])
            self._lin_grads(dx, f, b + "mlp.2.weight", acc, bias=((b + "mlp.2.bias", 0, d),))
            dz = ops.gelu_bwd(R["z"], ops.gemm_nt(dx, w[f"{i}.fc2T"]))
            del f
            self._lin_grads(dz, R["h2"], b + "mlp.0.weight", acc, bias=((b + "mlp.0.bias", 0, 4 * d),))
            dh2 = ops.gemm_nt(dz, w[f"{i}.fc1T"])
            del dz
            dx1 = ops.layernorm_bwd(R["x1"], R["m2"], R["r2"], w[f"{i}.ln2_w"], dh2, dgamma=gv(b + "mlp_ln.weight"),
                                    dbeta=gv(b + "mlp_ln.bias"), accumulate=acc)
            ops.add_(dx1, dx)
            # attention: x1 = x + out(attn(qkv(LN1(x))))
            self._lin_grads(dx1, R["a"], b + "attn.out.weight", acc, bias=((b + "attn.out.bias", 0, d),))
            da = ops.gemm_nt(dx1, w[f"{i}.outT"])
            qkv = R["qkv"]
            dqkv = torch.empty_like(qkv)
            self._attn_bwd_geom(qkv, R["a"], da, R["lse"], dqkv, geom)
            del da
            self._lin_grads(dqkv, R["h"], b + "attn.query.weight", acc, N=3 * d, K=d,
                            bias=((b + "attn.query.bias", 0, d), (b + "attn.value.bias", 2 * d, d)))
            dh = ops.gemm_nt(dqkv, w[f"{i}.qkvT"])
            del dqkv
            dx = ops.layernorm_bwd(R["x"], R["m1"], R["r1"], w[f"{i}.ln1_w"], dh, dgamma=gv(b + "attn_ln.weight"),
                                   dbeta=gv(b + "attn_ln.bias"), accumulate=acc)
            ops.add_(dx, dx1)
            S["blocks"][i] = None
        # conv stem: x0 = gelu(conv2(gelu(conv1(mel)))) + pos  (pos is a fixed buffer)
        if nf is not None:      # packed rows -> the padded conv2 layout (rows past a clip's own frames carry no gradient)
            dx = ops.gather_rows(dx, S["inv2"])
        dz2 = ops.gelu_bwd(S["z2"], dx)
        h1 = ops.gelu_fwd(S["z1"])
        cols2 = ops.conv1d_k3_im2col(h1.view(B, T, d), 2, 3 * d, n_valid=nf)
        del h1
        Mp = round_up(M, 64)
        g2 = ops.gemm_nt(ops.transpose(dz2, Rp=Mp), ops.transpose(cols2, Rp=Mp), out_dtype=torch.float32)   # [d, 3d], tap-major
        del cols2
        g2 = g2.view(d, 3, d).permute(0, 2, 1)          # -> the parameter's [co, ci, tap] layout
        gv(p + "conv2.weight").add_(g2) if acc else gv(p + "conv2.weight").copy_(g2)
        ops.colsum(dz2, gv(p + "conv2.bias"), accumulate=acc)
        dh1 = ops.conv1d_k3_col2im(ops.gemm_nt(dz2, w["conv2T"]), B, T, d, 2)
        if nf is not None:      # conv2 read ZEROS (not conv1's output) past a clip's own frames: no gradient flows to those conv1 rows
            dh1 = ops.gather_rows(dh1.view(B * T, d), S["valid1"]).view(B, T, d)
        dz1 = ops.gelu_bwd(S["z1"], dh1.view(B * T, d))
        cols1 = ops.conv1d_k3_im2col(S["mel"], 1, self.kp1, n_valid=nf)
        M1p = round_up(B * T, 64)
        g1 = ops.gemm_nt(ops.transpose(dz1, Rp=M1p), ops.transpose(cols1, Rp=M1p), out_dtype=torch.float32)  # [d, kp1]
        g1 = g1[:, : 3 * nm].reshape(d, 3, nm).permute(0, 2, 1)
        gv(p + "conv1.weight").add_(g1) if acc else gv(p + "conv1.weight").copy_(g1)
        ops.colsum(dz1, gv(p + "conv1.bias"), accumulate=acc)

    @torch.no_grad()
    def forward_packed(self, mel: torch.Tensor, n_frames: List[int]):
        """Ragged batch without pad frames (`++model_config.varlen_encoder=true`, pad_or_trim off).
        mel [B, Tmax, n_mels] f32, zero padded in mel space like the reference's collator (speech_dataset_large.py:194-197);
        n_frames[b] = real mel frames of clip b (host ints).  Every clip is encoded exactly as if it were alone in the
        batch (B = 1 through src/slam_llm/models/encoder.py:13-30): conv2 sees zeros past the clip's own conv1 frames, the
        positional embedding restarts at every clip, attention is restricted to the clip's own frames (seg_lo / seg_hi).
        This is a STATED deviation from the reference for ragged batches, where real frames also attend to the pad frames
        of the zero-padded batch; it is identical to the reference for B = 1 and for equal-length clips.
        Returns (packed encoder output [sum T2_b, d] bf16, T2 list)."""
        cfg, w = self.cfg, self.w
        B, T, nm = mel.shape
        d, H, dev = cfg["enc_dim"], cfg["enc_heads"], mel.device
        assert len(n_frames) == B and max(n_frames) <= T
        T2 = [(n + 1) // 2 for n in n_frames]
        T2max = (T + 1) // 2
        assert max(T2) <= w["pos"].shape[0], "audio longer than the encoder's positional table"
        nf = torch.tensor(n_frames, dtype=torch.int32).to(dev, non_blocking=True)
        cols = ops.conv1d_k3_im2col(mel, 1, self.kp1, n_valid=nf)
        h1 = ops.gemm_nt(cols, w["conv1"], bias=w["conv1_b"], act=ACT_GELU)
        del cols
        cols2 = ops.conv1d_k3_im2col(h1.view(B, T, d), 2, 3 * d, n_valid=nf)   # rows past a clip's conv1 frames read as zero
        xpad = ops.gemm_nt(cols2, w["conv2"], bias=w["conv2_b"], act=ACT_GELU, residual=w["pos"], res_row_mod=T2max)
        del cols2, h1
        # pack the valid rows; per-row segment bounds for the attention
        starts, idx, lo, hi = [], [], [], []
        acc = 0
        for b_, t2 in enumerate(T2):
            starts.append(acc)
            idx.append(torch.arange(b_ * T2max, b_ * T2max + t2, dtype=torch.int32))
            lo.append(torch.full((t2,), acc, dtype=torch.int32))
            hi.append(torch.full((t2,), acc + t2, dtype=torch.int32))
            acc += t2
        M = acc
        meta = torch.stack([torch.cat(idx), torch.cat(lo), torch.cat(hi)]).to(dev, non_blocking=True)
        x = ops.gather_rows(xpad, meta[0])
        del xpad
        seg = (meta[1], meta[2])
        scale = 64 ** -0.5
        hbuf = torch.empty((M, d), dtype=torch.bfloat16, device=dev)
        qkv = torch.empty((M, 3 * d), dtype=torch.bfloat16, device=dev)
        obuf = torch.empty((M, d), dtype=torch.bfloat16, device=dev)
        fbuf = torch.empty((M, 4 * d), dtype=torch.bfloat16, device=dev)
        for i in range(cfg["enc_layers"]):
            ops.layernorm(x, w[f"{i}.ln1_w"], w[f"{i}.ln1_b"], out=hbuf)
            ops.gemm_nt(hbuf, w[f"{i}.qkv"], out=qkv, bias=w[f"{i}.qkv_b"])
            ops.attn_fwd(qkv[:, :d], qkv[:, d: 2 * d], qkv[:, 2 * d:], 1, M, H, H, 64, False, scale, want_lse=False, out=obuf, seg=seg, q_prescaled=not self.trainable)
            ops.gemm_nt(obuf, w[f"{i}.out"], out=x, bias=w[f"{i}.out_b"], residual=x)
            ops.layernorm(x, w[f"{i}.ln2_w"], w[f"{i}.ln2_b"], out=hbuf)
            ops.gemm_nt(hbuf, w[f"{i}.fc1"], out=fbuf, bias=w[f"{i}.fc1_b"], act=ACT_GELU)
            ops.gemm_nt(fbuf, w[f"{i}.fc2"], out=x, bias=w[f"{i}.fc2_b"], residual=x)
        out = ops.layernorm(x, w["lnp_w"], w["lnp_b"])
        return out, T2



def conv_stack_pitches(n_samples: int, kernels, strides):
    """valid rows T_i of every conv layer (no padding) and the smallest per-clip row pitches P_i >= T_i with P_i = stride_(i+1) * P_(i+1):
    on those pitches row b * P_(i+1) + t of layer i + 1 starts at row stride * (b * P_(i+1) + t) of layer i's buffer for every clip"""
    L = len(kernels)
    Ts = [(n_samples - kernels[0]) // strides[0] + 1]
    for i in range(1, L):
        Ts.append((Ts[-1] - kernels[i]) // strides[i] + 1)
    P_last = Ts[-1]
    while True:
        P = [0] * L
        P[-1] = P_last
        for i in range(L - 2, -1, -1):
            P[i] = P[i + 1] * strides[i + 1]
        if all(P[i] >= Ts[i] for i in range(L)):
            return Ts, P
        P_last += 1


# ======================================================================================== hubert encoder
class HipHubertEncoder(nn.Module):
    """Frozen HuBERT encoder (fairseq HubertModel as called at src/slam_llm/models/slam_model.py:335-341;
    architecture per its HF twin, transformers/models/hubert/modeling_hubert.py).  Base geometry (`hub_extractor_mode="default"`,
    `hub_layer_norm_first=False`): GroupNorm over time after the first conv only, conv -> GELU for the rest, post-LN layers behind
    the encoder LayerNorm.  Large / xlarge: 7 LayerNorm conv layers
    (im2col + MFMA GEMM + fused LayerNorm-GELU), feature projection, grouped positional conv (one GEMM per group with
    the GELU and the residual add fused in the epilogue), pre-LN transformer, final LayerNorm.  Inference-only.
    Weights use the HF state-dict names under `encoder.` (fairseq -> HF renaming is HF's conversion script)."""

    def __init__(self, cfg: dict, device, store: Optional["TrainableStore"] = None, prefix="encoder."):
        super().__init__()
        self._drop_calls = 0
        self.cfg, self.device_, self.w = cfg, device, {}
        self.store, self.prefix = store, prefix
        if store is not None:
            self._reserve_trainable()

    # ---- trainable form (train_config.freeze_encoder=false, models/slam_model.py:110-113) ------------------------------------
    @property
    def trainable(self) -> bool:
        return self.store is not None

    @property
    def _group_mode(self) -> bool:
        """base geometries: conv -> GroupNorm over time -> GELU on layer 0, conv -> GELU after it (extractor_mode "default")"""
        return self.cfg.get("hub_extractor_mode", "layer_norm") == "default"

    @property
    def _pre_ln(self) -> bool:
        return bool(self.cfg.get("hub_layer_norm_first", True))

    def _regularisers(self) -> SimpleNamespace:
        """the train-mode regularisers of the UN-FROZEN encoder (the reference leaves it in train mode, slam_model.py:317-318 calls .eval()
        only when frozen): WavLM.py:180-185 / fairseq HubertConfig -- `dropout` (after the positional conv, WavLM.py:584, and after the
        attention / feed-forward output projections, :702-726 dropout1 / dropout3), `attention_dropout` (on the attention probabilities),
        `activation_dropout` (after the GELU, dropout2), `dropout_input` (after post_extract_proj, :353) and `encoder_layerdrop` (:597).
        cfg keys hub_dropout / hub_attention_dropout / hub_activation_dropout / hub_dropout_input / hub_layerdrop; absent = 0 (plain
        geometry configs), the factory fills in the reference's defaults (slam_model_hip.build_config).  All zero in eval mode or when
        the encoder is frozen."""
        c, on = self.cfg, bool(self.trainable and self.training)
        f = (lambda k: float(c.get(k, 0.0) or 0.0)) if on else (lambda k: 0.0)
        R = SimpleNamespace(p=f("hub_dropout"), p_attn=f("hub_attention_dropout"), p_act=f("hub_activation_dropout"),
                            p_in=f("hub_dropout_input"), layerdrop=f("hub_layerdrop"))
        R.on = on and max(R.p, R.p_attn, R.p_act, R.p_in, R.layerdrop) > 0.0
        seed = (torch.initial_seed() ^ 0xE7C0DE) if R.on else 0

        def key(p):
            """(p, seed, offset) of the next hidden dropout (slam_dropout_bf16's counter-based mask; the backward recomputes it)"""
            if not (R.on and p > 0.0):
                return None
            self._drop_calls += 1
            # the offset carries 20 bits of the call counter (an encoder draws ~100 keys per step: 2^20 keys = ~10 k steps); the bits above
            # go into the seed, so a long run never re-uses a (seed, offset) pair (the counter used to wrap onto old masks)
            epoch = self._drop_calls >> 20
            s = seed if epoch == 0 else (seed ^ (epoch * 0x9E3779B97F4A7C15)) & (2 ** 64 - 1)
            return (p, s, ((self._drop_calls & 0xFFFFF) | 0x100000) << 40)

        def attn_key(p):
            """(p, seed) of the next attention-probability dropout (applied inside the attention kernels)"""
            if not (R.on and p > 0.0):
                return None
            self._drop_calls += 1
            return (p, (seed * 0x9E3779B1 + self._drop_calls) & (2 ** 63 - 1))

        def keep_layer():
            """encoder_layerdrop: `np.random.random() > layerdrop` keeps the layer (WavLM.py:596-597: numpy's global stream, one draw per
            layer, drawn in train mode only here)"""
            return (not R.on) or bool(np.random.random() > R.layerdrop)
        R.key, R.attn_key, R.keep_layer = key, attn_key, keep_layer
        return R

    @staticmethod
    def _drop_add(t: torch.Tensor, residual: torch.Tensor, key) -> torch.Tensor:
        """residual + dropout(t) (the output projections' dropout1 / dropout3 sit BEFORE the residual add)"""
        return ops.dropout(t, *key, out=residual.clone(), accumulate=True)

    def _nm(self) -> SimpleNamespace:
        """logical tensor -> state-dict name of the trainable form.  Default: the names of the module the reference un-freezes -- fairseq's
        HubertModel (models/slam_model.py:110-113 sets requires_grad on `self.encoder`, built at models/encoder.py:130-140 by fairseq's
        checkpoint_utils), whose parameters are `encoder.feature_extractor.conv_layers.N.0.weight`, `encoder.post_extract_proj.*`,
        `encoder.encoder.pos_conv.0.weight_g / weight_v`, `encoder.encoder.layers.N.self_attn.q_proj.*`, ... (the module tree that
        src/slam_llm/models/wavlm/WavLM.py:220-330 vendors a copy of) -- so `named_parameters()` / checkpoints written from it carry the
        reference's keys and AdamW acts on (g, v) of the weight-normed positional conv like it does there.  `hub_param_names="hf"`: the HF
        twin's names with the positional conv FOLDED (w = g v / ||v||), for checkpoints that only exist in HF form."""
        if self.cfg.get("hub_param_names", "fairseq") == "hf":
            return self._nm_hf()
        return self._nm_fairseq()

    def _nm_fairseq(self) -> SimpleNamespace:
        p = self.prefix
        e = p + "encoder."
        c = lambda i: f"{p}feature_extractor.conv_layers.{i}."   # noqa: E731
        lyr = lambda i: f"{e}layers.{i}."                        # noqa: E731
        gm = self._group_mode     # "default" extractor: Fp32GroupNorm at index 2 of layer 0 only, no conv bias
        return SimpleNamespace(
            conv_w=lambda i: c(i) + "0.weight", conv_b=None if (gm or not self.cfg.get("hub_conv_bias", True)) else (lambda i: c(i) + "0.bias"),
            conv_ln=lambda i: (c(i) + "2" if i == 0 else None) if gm else c(i) + "2.1",
            fp_ln=p + "layer_norm", fp=p + "post_extract_proj",
            pos_w=None, pos_g=e + "pos_conv.0.weight_g", pos_v=e + "pos_conv.0.weight_v", pos_b=e + "pos_conv.0.bias",
            q=lambda i: lyr(i) + "self_attn.q_proj", k=lambda i: lyr(i) + "self_attn.k_proj", v=lambda i: lyr(i) + "self_attn.v_proj",
            out=lambda i: lyr(i) + "self_attn.out_proj", ln1=lambda i: lyr(i) + "self_attn_layer_norm",
            fc1=lambda i: lyr(i) + "fc1", fc2=lambda i: lyr(i) + "fc2", ln2=lambda i: lyr(i) + "final_layer_norm", enc_ln=e + "layer_norm")

    def _nm_hf(self) -> SimpleNamespace:
        p = self.prefix
        e = p + "encoder."
        c = lambda i: f"{p}feature_extractor.conv_layers.{i}."   # noqa: E731
        lyr = lambda i: f"{e}layers.{i}."                        # noqa: E731
        gm = self._group_mode     # base: GroupNorm on conv layer 0 only (HF names it layer_norm too), no conv bias
        return SimpleNamespace(
            conv_w=lambda i: c(i) + "conv.weight", conv_b=None if gm else (lambda i: c(i) + "conv.bias"),
            conv_ln=lambda i: None if (gm and i > 0) else c(i) + "layer_norm",
            fp_ln=p + "feature_projection.layer_norm", fp=p + "feature_projection.projection",
            pos_w=e + "pos_conv_embed.conv.weight", pos_g=None, pos_v=None, pos_b=e + "pos_conv_embed.conv.bias",
            q=lambda i: lyr(i) + "attention.q_proj", k=lambda i: lyr(i) + "attention.k_proj", v=lambda i: lyr(i) + "attention.v_proj",
            out=lambda i: lyr(i) + "attention.out_proj", ln1=lambda i: lyr(i) + "layer_norm",
            fc1=lambda i: lyr(i) + "feed_forward.intermediate_dense", fc2=lambda i: lyr(i) + "feed_forward.output_dense",
            ln2=lambda i: lyr(i) + "final_layer_norm", enc_ln=e + "layer_norm")

    def key_map(self, src_style: str = "hf") -> Dict[str, str]:
        """{name in this module's trainable store: name of the same tensor under `src_style` ("hf" | "fairseq")} for every tensor both
        styles hold under their own key (everything but the positional conv's weight, which is (weight_g, weight_v) in fairseq form and
        one folded `conv.weight` -- or `parametrizations.weight.original0 / original1` -- in HF form; `_load_trainable` converts it)."""
        mine, other = self._nm(), (self._nm_hf() if src_style == "hf" else self._nm_fairseq())
        cfg, out = self.cfg, {}

        def pair(a, b, suffixes=(".weight", ".bias")):
            if a is None or b is None:
                return
            for sfx in suffixes:
                out[a + sfx] = b + sfx
        for i in range(len(cfg["hub_conv_dim"])):
            pair(mine.conv_w(i), other.conv_w(i), ("",))
            if mine.conv_b is not None and other.conv_b is not None:
                pair(mine.conv_b(i), other.conv_b(i), ("",))
            pair(mine.conv_ln(i), other.conv_ln(i))
        pair(mine.fp_ln, other.fp_ln); pair(mine.fp, other.fp); pair(mine.enc_ln, other.enc_ln)
        pair(mine.pos_b, other.pos_b, ("",))
        for i in range(cfg["hub_layers"]):
            for f in ("q", "k", "v", "out", "ln1", "fc1", "fc2", "ln2"):
                pair(getattr(mine, f)(i), getattr(other, f)(i))
        return out

    def _reserve_extra_layer(self, i: int):
        """hook: further trainable tensors of layer i (WavLM: the gate of the relative position bias)"""

    def _reserve_extra_top(self):
        """hook: further trainable tensors outside the layers (WavLM: mask_emb)"""

    def _reserve_trainable(self):
        """every parameter of the encoder joins the flat fp32 store under its state-dict name, in the order the backward produces
        the gradients (GradSync prefixes): final LayerNorm, layers last -> first, positional conv, feature projection, conv stack
        last -> first (post-LN geometries produce the encoder LayerNorm's after the layers; the encoder region is flushed to the
        gradient hooks as a whole, so only the LLM prefix order matters).  Both graphs: "layer_norm" extractor + layer_norm_first layers
        (large / xlarge / WavLM-Large) and the base geometries (GroupNorm-over-time extractor, post-LN layers).  HuBERT (HF names): the weight-normed positional conv trains its FOLDED weight (w = g v / ||v|| formed once at load);
        WavLM (reference names): weight_g / weight_v stay the parameters, the fold and its chain rule run every step."""
        cfg, r, N = self.cfg, self.store.reserve, self._nm()
        d, Fd = cfg["hub_dim"], cfg["hub_ffn"]
        assert d % 64 == 0 and d // cfg["hub_heads"] == 64
        r(N.enc_ln + ".weight", (d,)); r(N.enc_ln + ".bias", (d,))
        for i in reversed(range(cfg["hub_layers"])):
            r(N.fc2(i) + ".weight", (d, Fd)); r(N.fc2(i) + ".bias", (d,))
            r(N.fc1(i) + ".weight", (Fd, d)); r(N.fc1(i) + ".bias", (Fd,))
            r(N.ln2(i) + ".weight", (d,)); r(N.ln2(i) + ".bias", (d,))
            r(N.out(i) + ".weight", (d, d)); r(N.out(i) + ".bias", (d,))
            for n in (N.q, N.k, N.v):     # back to back: the bf16 copies form the fused [3d, d] operand
                r(n(i) + ".weight", (d, d))
            for n in (N.q, N.k, N.v):     # ... and the fused [3d] bias
                r(n(i) + ".bias", (d,))
            self._reserve_extra_layer(i)
            r(N.ln1(i) + ".weight", (d,)); r(N.ln1(i) + ".bias", (d,))
        gch, kpos = d // cfg["hub_pos_groups"], cfg["hub_pos_k"]
        if N.pos_w is not None:
            r(N.pos_w, (d, gch, kpos))
        else:
            r(N.pos_g, (1, 1, kpos)); r(N.pos_v, (d, gch, kpos))
        r(N.pos_b, (d,))
        cin = cfg["hub_conv_dim"][-1]
        r(N.fp + ".weight", (d, cin)); r(N.fp + ".bias", (d,))
        r(N.fp_ln + ".weight", (cin,)); r(N.fp_ln + ".bias", (cin,))
        dims = [1] + list(cfg["hub_conv_dim"])
        for i in reversed(range(len(cfg["hub_conv_dim"]))):
            if N.conv_ln(i) is not None:
                r(N.conv_ln(i) + ".weight", (dims[i + 1],)); r(N.conv_ln(i) + ".bias", (dims[i + 1],))
            r(N.conv_w(i), (dims[i + 1], dims[i], cfg["hub_conv_kernel"][i]))
            if N.conv_b is not None:
                r(N.conv_b(i), (dims[i + 1],))
        self._reserve_extra_top()

    def bind(self):
        # this module is SlamHipModel.encoder: the registered path below it is the name minus "encoder." (WavLM keeps its "model." level)
        for name, prm in self.store.params.items():
            if name.startswith(self.prefix):
                _attach(self, name[len("encoder."):], prm)

    def _pos_conv_sources(self, W: Dict[str, torch.Tensor], prefix: str):
        """(g, v, folded) of the weight-normed positional conv from whichever form the checkpoint holds: fairseq `weight_g / weight_v`,
        torch's parametrizations API (`parametrizations.weight.original0 / original1`, fairseq or HF module path), or HF's folded
        `conv.weight` -- for which (g, v) = (||w||, w) is the parametrisation of the same function (w = g v / ||v||)."""
        rel = lambda n: prefix + n[len(self.prefix):]   # noqa: E731
        F_, H_ = self._nm_fairseq(), self._nm_hf()
        fb, hb = rel(F_.pos_g)[: -len("weight_g")], rel(H_.pos_w)[: -len("weight")]
        for base in (fb, hb):
            if base + "weight_g" in W:
                return W[base + "weight_g"].float(), W[base + "weight_v"].float(), None
            if base + "parametrizations.weight.original0" in W:
                return W[base + "parametrizations.weight.original0"].float(), W[base + "parametrizations.weight.original1"].float(), None
        w = W[hb + "weight"].float() if hb + "weight" in W else W[fb + "weight"].float()
        return w.norm(dim=(0, 1), keepdim=True), w, w

    def _load_trainable(self, W: Dict[str, torch.Tensor], prefix: str):
        """checkpoint -> store.  Keys may be in the store's own style or in the other one (`key_map`): the reference's fairseq checkpoint
        and an HF-converted one both load."""
        N = self._nm()
        alt = self.key_map("hf" if self.cfg.get("hub_param_names", "fairseq") != "hf" else "fairseq") if type(self)._nm is HipHubertEncoder._nm else {}
        with torch.no_grad():
            for name, prm in self.store.params.items():
                if not name.startswith(self.prefix):
                    continue
                src = prefix + name[len(self.prefix):]
                if name in (N.pos_w, N.pos_g, N.pos_v) and src not in W:
                    g_, v_, folded = self._pos_conv_sources(W, prefix)
                    val = {N.pos_g: g_, N.pos_v: v_}.get(name) if name != N.pos_w else (folded if folded is not None else g_ * v_ / v_.norm(dim=(0, 1), keepdim=True))
                    prm.copy_(val.to(self.device_).reshape(prm.shape))
                    continue
                if src not in W and name in alt:
                    src = prefix + alt[name][len(self.prefix):]
                prm.copy_(W[src].to(self.device_).reshape(prm.shape))
        return self

    def _pos_weight_master(self) -> torch.Tensor:
        """fp32 [d, gch, kpos] weight of the positional conv as the forward uses it"""
        N, st = self._nm(), self.store
        if N.pos_w is not None:
            return st.master_view(N.pos_w)
        g_, v_ = st.master_view(N.pos_g), st.master_view(N.pos_v)      # nn.utils.weight_norm(dim=2): w = g * v / ||v||_(0,1)
        return g_ * v_ / v_.norm(dim=(0, 1), keepdim=True)

    def refresh(self):
        """rebuild the bf16 compute copies (and the transposes / packings the backward multiplies by) from the store"""
        if not self.trainable:
            return
        st, cfg, w, dev, N = self.store, self.cfg, self.w, self.device_, self._nm()
        d = cfg["hub_dim"]
        cin = 1
        for i, (co, k) in enumerate(zip(cfg["hub_conv_dim"], cfg["hub_conv_kernel"])):
            kp = round_up(k * cin, 64)
            wc = torch.zeros((co, kp), dtype=torch.bfloat16, device=dev)
            wc[:, : k * cin] = st.bf16_view(N.conv_w(i)).permute(0, 2, 1).reshape(co, k * cin)   # tap-major columns (im2col order)
            w[f"c{i}"] = wc
            w[f"c{i}_b"] = st.master_view(N.conv_b(i)) if N.conv_b is not None else torch.zeros(co, dtype=torch.float32, device=dev)
            w[f"c{i}T"] = ops.transpose(wc, Rp=co)                                             # [kp, co]: dcols = dy . Wc
            if N.conv_ln(i) is not None:
                w[f"c{i}_lw"], w[f"c{i}_lb"] = st.master_view(N.conv_ln(i) + ".weight"), st.master_view(N.conv_ln(i) + ".bias")
            cin = co
        w["fp_lw"], w["fp_lb"] = st.master_view(N.fp_ln + ".weight"), st.master_view(N.fp_ln + ".bias")
        w["fp"], w["fp_b"] = st.bf16_view(N.fp + ".weight"), st.master_view(N.fp + ".bias")
        w["fpT"] = ops.transpose(w["fp"], Rp=d)
        G, kpos = cfg["hub_pos_groups"], cfg["hub_pos_k"]
        gch = d // G
        self.pos_kp = round_up(kpos * gch, 64)
        pg = torch.zeros((G, gch, self.pos_kp), dtype=torch.bfloat16, device=dev)
        pg[:, :, : kpos * gch] = self._pos_weight_master().to(torch.bfloat16).view(G, gch, gch, kpos).permute(0, 1, 3, 2).reshape(G, gch, kpos * gch)
        w["pos"], w["pos_b"] = pg, st.master_view(N.pos_b)
        w["pos_tap"] = ops.pos_conv_pack(pg, kpos)
        w["pos_adj"] = ops.pos_conv_pack_adjoint(pg, kpos)
        for i in range(cfg["hub_layers"]):
            off = st.offsets[N.q(i) + ".weight"][0]
            w[f"{i}.qkv"] = st.flat_bf16[off: off + 3 * d * d].view(3 * d, d)
            boff = st.offsets[N.q(i) + ".bias"][0]
            w[f"{i}.qkv_b"] = st.flat[boff: boff + 3 * d]
            w[f"{i}.out"], w[f"{i}.out_b"] = st.bf16_view(N.out(i) + ".weight"), st.master_view(N.out(i) + ".bias")
            w[f"{i}.ln1_w"], w[f"{i}.ln1_b"] = st.master_view(N.ln1(i) + ".weight"), st.master_view(N.ln1(i) + ".bias")
            w[f"{i}.fc1"], w[f"{i}.fc1_b"] = st.bf16_view(N.fc1(i) + ".weight"), st.master_view(N.fc1(i) + ".bias")
            w[f"{i}.fc2"], w[f"{i}.fc2_b"] = st.bf16_view(N.fc2(i) + ".weight"), st.master_view(N.fc2(i) + ".bias")
            w[f"{i}.ln2_w"], w[f"{i}.ln2_b"] = st.master_view(N.ln2(i) + ".weight"), st.master_view(N.ln2(i) + ".bias")
            for nme in ("qkv", "out", "fc1", "fc2"):
                w[f"{i}.{nme}T"] = ops.transpose(w[f"{i}.{nme}"], Rp=w[f"{i}.{nme}"].shape[0])
        w["lnp_w"], w["lnp_b"] = st.master_view(N.enc_ln + ".weight"), st.master_view(N.enc_ln + ".bias")
        self._refresh_extra()

    def _refresh_extra(self):
        """hook: compute copies of the extra trainable tensors (WavLM)"""

    def load(self, W: Dict[str, torch.Tensor], prefix="encoder."):
        if self.trainable:
            return self._load_trainable(W, prefix)
        cfg, dev, w = self.cfg, self.device_, self.w
        bf = lambda t: t.to(device=dev, dtype=torch.bfloat16).contiguous()  # noqa: E731
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        cin = 1
        for i, (co, k) in enumerate(zip(cfg["hub_conv_dim"], cfg["hub_conv_kernel"])):
            p = f"{prefix}feature_extractor.conv_layers.{i}."
            kp = round_up(k * cin, 64)
            wc = torch.zeros((co, kp), dtype=torch.bfloat16, device=dev)
            wc[:, : k * cin] = bf(W[p + "conv.weight"].permute(0, 2, 1).reshape(co, k * cin))
            cb = W.get(p + "conv.bias")                                  # conv_bias=False in the base configuration
            w[f"c{i}"], w[f"c{i}_b"] = wc, (f32(cb) if cb is not None else torch.zeros(co, dtype=torch.float32, device=dev))
            if p + "layer_norm.weight" in W:                             # base ("default" extractor): GroupNorm on layer 0 only
                w[f"c{i}_lw"], w[f"c{i}_lb"] = f32(W[p + "layer_norm.weight"]), f32(W[p + "layer_norm.bias"])
            cin = co
        d = cfg["hub_dim"]
        assert d % 64 == 0 and d // cfg["hub_heads"] == 64 and cin % 64 == 0
        p = prefix + "feature_projection."
        w["fp_lw"], w["fp_lb"] = f32(W[p + "layer_norm.weight"]), f32(W[p + "layer_norm.bias"])
        w["fp"], w["fp_b"] = bf(W[p + "projection.weight"]), f32(W[p + "projection.bias"])
        p = prefix + "encoder."
        key = p + "pos_conv_embed.conv.weight"
        if key in W:
            pw = W[key].float()
        else:  # HF checkpoints keep the weight-norm parametrisation (dim=2): w = g * v / ||v||
            g_, v_ = W[p + "pos_conv_embed.conv.parametrizations.weight.original0"].float(), W[p + "pos_conv_embed.conv.parametrizations.weight.original1"].float()
            pw = g_ * v_ / v_.norm(dim=(0, 1), keepdim=True)
        G, kpos = cfg["hub_pos_groups"], cfg["hub_pos_k"]
        gch = d // G
        self.pos_kp = round_up(kpos * gch, 64)
        pg = torch.zeros((G, gch, self.pos_kp), dtype=torch.bfloat16, device=dev)
        for g in range(G):
            pg[g, :, : kpos * gch] = bf(pw[g * gch:(g + 1) * gch].permute(0, 2, 1).reshape(gch, kpos * gch))
        w["pos"], w["pos_b"] = pg, f32(W[p + "pos_conv_embed.conv.bias"])
        for i in range(cfg["hub_layers"]):
            q = f"{p}layers.{i}."
            w[f"{i}.qkv"] = bf(torch.cat([W[q + "attention.q_proj.weight"], W[q + "attention.k_proj.weight"], W[q + "attention.v_proj.weight"]], 0))
            w[f"{i}.qkv_b"] = f32(torch.cat([W[q + "attention.q_proj.bias"], W[q + "attention.k_proj.bias"], W[q + "attention.v_proj.bias"]], 0))
            w[f"{i}.out"], w[f"{i}.out_b"] = bf(W[q + "attention.out_proj.weight"]), f32(W[q + "attention.out_proj.bias"])
            w[f"{i}.ln1_w"], w[f"{i}.ln1_b"] = f32(W[q + "layer_norm.weight"]), f32(W[q + "layer_norm.bias"])
            w[f"{i}.fc1"], w[f"{i}.fc1_b"] = bf(W[q + "feed_forward.intermediate_dense.weight"]), f32(W[q + "feed_forward.intermediate_dense.bias"])
            w[f"{i}.fc2"], w[f"{i}.fc2_b"] = bf(W[q + "feed_forward.output_dense.weight"]), f32(W[q + "feed_forward.output_dense.bias"])
            w[f"{i}.ln2_w"], w[f"{i}.ln2_b"] = f32(W[q + "final_layer_norm.weight"]), f32(W[q + "final_layer_norm.bias"])
        w["lnp_w"], w["lnp_b"] = f32(W[p + "layer_norm.weight"]), f32(W[p + "layer_norm.bias"])
        return self

    def init_random(self, seed: int = 42):
        """seeded random weights generated directly in HBM at the true dimensions (benchmarks: no checkpoints offline)"""
        cfg, dev, w = self.cfg, self.device_, self.w
        g = torch.Generator(device=dev).manual_seed(seed)
        rn = lambda *s, std=0.02: (torch.randn(*s, generator=g, device=dev) * std)  # noqa: E731
        if self.trainable:      # the parameters live in the store: LayerNorm 1 / small, biases small, weights ~ fan_in^-1/2
            with torch.no_grad():
                for name, prm in self.store.params.items():
                    if not name.startswith(self.prefix):
                        continue
                    if "layer_norm" in name:
                        prm.copy_((1 + rn(*prm.shape, std=0.1)) if name.endswith("weight") else rn(*prm.shape, std=0.1))
                    elif name.endswith("bias"):
                        prm.copy_(rn(*prm.shape))
                    else:
                        prm.copy_(rn(*prm.shape, std=float(prm[0].numel()) ** -0.5))
            return self
        cin = 1
        for i, (co, k) in enumerate(zip(cfg["hub_conv_dim"], cfg["hub_conv_kernel"])):
            kp = round_up(k * cin, 64)
            wc = torch.zeros((co, kp), dtype=torch.bfloat16, device=dev)
            wc[:, : k * cin] = rn(co, k * cin, std=(k * cin) ** -0.5).to(torch.bfloat16)
            w[f"c{i}"], w[f"c{i}_b"] = wc, rn(co)
            w[f"c{i}_lw"], w[f"c{i}_lb"] = 1 + rn(co, std=0.1), rn(co, std=0.1)
            cin = co
        d, Fd = cfg["hub_dim"], cfg["hub_ffn"]
        assert d % 64 == 0 and d // cfg["hub_heads"] == 64 and cin % 64 == 0
        w["fp_lw"], w["fp_lb"] = 1 + rn(cin, std=0.1), rn(cin, std=0.1)
        w["fp"], w["fp_b"] = rn(d, cin, std=cin ** -0.5).to(torch.bfloat16), rn(d)
        G, kpos = cfg["hub_pos_groups"], cfg["hub_pos_k"]
        gch = d // G
        self.pos_kp = round_up(kpos * gch, 64)
        pg = torch.zeros((G, gch, self.pos_kp), dtype=torch.bfloat16, device=dev)
        pg[:, :, : kpos * gch] = rn(G, gch, kpos * gch, std=(kpos * gch) ** -0.5).to(torch.bfloat16)
        w["pos"], w["pos_b"] = pg, rn(d)
        for i in range(cfg["hub_layers"]):
            w[f"{i}.qkv"], w[f"{i}.qkv_b"] = rn(3 * d, d, std=d ** -0.5).to(torch.bfloat16), rn(3 * d)
            w[f"{i}.out"], w[f"{i}.out_b"] = rn(d, d, std=d ** -0.5).to(torch.bfloat16), rn(d)
            w[f"{i}.ln1_w"], w[f"{i}.ln1_b"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
            w[f"{i}.fc1"], w[f"{i}.fc1_b"] = rn(Fd, d, std=d ** -0.5).to(torch.bfloat16), rn(Fd)
            w[f"{i}.fc2"], w[f"{i}.fc2_b"] = rn(d, Fd, std=Fd ** -0.5).to(torch.bfloat16), rn(d)
            w[f"{i}.ln2_w"], w[f"{i}.ln2_b"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
        w["lnp_w"], w["lnp_b"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
        return self

    def _conv_by_view(self, i: int, cin: int, k: int) -> bool:
        return CONV_WINDOW_VIEW and i > 0 and (k * cin) % 64 == 0 and self.w[f"c{i}"].shape[1] == k * cin

    def _conv_view_gemm(self, x2d: torch.Tensor, B: int, Tin: int, cin: int, k: int, st: int, i: int, act=ACT_NONE) -> torch.Tensor:
        """conv layer i as B GEMMs over the overlapping-row window view of the [B * Tin, cin] signal (see CONV_WINDOW_VIEW)"""
        w = self.w
        Tout = (Tin - k) // st + 1
        assert x2d.is_contiguous() and x2d.shape == (B * Tin, cin)
        y = torch.empty((B * Tout, w[f"c{i}"].shape[0]), dtype=torch.bfloat16, device=x2d.device)
        for b_ in range(B):
            a = x2d.as_strided((Tout, k * cin), (st * cin, 1), x2d.storage_offset() + b_ * Tin * cin)
            ops.gemm_nt(a, w[f"c{i}"], out=y[b_ * Tout:(b_ + 1) * Tout], bias=w[f"c{i}_b"], act=act)
        return y

    def _conv_stack_pitched(self, wav2d: torch.Tensor, B: int, N: int):
        """the "layer_norm" conv stack (inference) with ONE GEMM launch per layer over the window view: layer i's rows live at a per-clip
        pitch P_i >= T_i chosen so that P_i = stride_{i+1} * P_{i+1} -- then row r = b * P_{i+1} + t of layer i + 1 reads the k * C contiguous
        elements from row stride * r of layer i's buffer for EVERY clip (uniform lda), the P_i - T_i rows at the end of a clip are computed from
        whatever follows and never read by a valid row (valid row t of layer i + 1 ends at input row T_i - 1 or before).  HuBERT-large,
        30 s: pitches 96000 / 48000 / ... / 1500 for 95999 / 47999 / ... / 1499 valid rows.  Returns (features [B * T_last, C], T_last, C)
        or None when a layer cannot take the view (tiny channel counts)."""
        cfg, w = self.cfg, self.w
        ks, ss, dims = list(cfg["hub_conv_kernel"]), list(cfg["hub_conv_stride"]), list(cfg["hub_conv_dim"])
        L = len(dims)
        if not CONV_WINDOW_VIEW or any(not self._conv_by_view(i, dims[i - 1], ks[i]) for i in range(1, L)):
            return None
        Ts, P = conv_stack_pitches(N, ks, ss)
        dev = wav2d.device
        slack = max(ks)
        # layer 0 (cin = 1, k = 10: its 10-wide windows do go through im2col), LayerNorm + GELU written clip by clip at pitch P[0]
        cols, T0 = ops.conv1d_im2col(wav2d, B, N, 0, 1, ks[0], ss[0], 0, Kp=w["c0"].shape[1])
        y = ops.gemm_nt(cols, w["c0"], bias=w["c0_b"])
        del cols
        buf = torch.empty((B * P[0] + slack, dims[0]), dtype=torch.bfloat16, device=dev)
        buf[B * P[0]:].zero_()
        if P[0] > T0:
            buf[: B * P[0]].view(B, P[0], dims[0])[:, T0:].zero_()
        for b_ in range(B):
            ops.layernorm(y[b_ * T0:(b_ + 1) * T0], w["c0_lw"], w["c0_lb"], 1e-5, out=buf[b_ * P[0]: b_ * P[0] + T0], gelu=True)
        del y
        for i in range(1, L):
            cin, co = dims[i - 1], dims[i]
            a = buf.as_strided((B * P[i], ks[i] * cin), (ss[i] * cin, 1), buf.storage_offset())
            nxt = torch.empty((B * P[i] + slack, co), dtype=torch.bfloat16, device=dev)
            nxt[B * P[i]:].zero_()
            ops.gemm_nt(a, w[f"c{i}"], out=nxt[: B * P[i]], bias=w[f"c{i}_b"])
            ops.layernorm(nxt[: B * P[i]], w[f"c{i}_lw"], w[f"c{i}_lb"], 1e-5, out=nxt[: B * P[i]], gelu=True)
            buf = nxt
        key = (B, P[-1], Ts[-1])
        if getattr(self, "_pitch_idx_key", None) != key:
            idx = (torch.arange(B, dtype=torch.int32)[:, None] * P[-1] + torch.arange(Ts[-1], dtype=torch.int32)[None, :]).reshape(-1)
            self._pitch_idx, self._pitch_idx_key = idx.to(dev), key
        feats = ops.gather_rows(buf[: B * P[-1]], self._pitch_idx) if P[-1] > Ts[-1] else buf[: B * P[-1]]
        return feats, Ts[-1], dims[-1]

    def out_frames(self, n: int) -> int:
        for k, s_ in zip(self.cfg["hub_conv_kernel"], self.cfg["hub_conv_stride"]):
            n = (n - k) // s_ + 1
        return n

    def valid_frames(self, n_padded: int, n_valid: List[int]) -> List[int]:
        """frames fairseq keeps per clip: its forward_padding_mask views the sample mask as [B, T', N // T'] and calls a frame
        padding iff ALL its samples are (fairseq/models/hubert/hubert.py; the reference passes `padding_mask = 1 - audio_mask`,
        slam_model.py:336) -> ceil(n / (N // T')), at most T'."""
        T = self.out_frames(n_padded)
        chunk = n_padded // T
        return [min(T, (int(n) + chunk - 1) // chunk) for n in n_valid]

    @torch.no_grad()
    def forward_wav(self, wav: torch.Tensor, n_valid: Optional[List[int]] = None) -> torch.Tensor:
        """wav [B, N] f32 (already normalised by the dataset) -> [B, T', hub_dim] bf16.
        n_valid (host ints, ragged batch of zero-padded waveforms): the conv stack runs over the padded waveform like the
        reference's, padded frames (fairseq's rule, valid_frames) are zeroed before the positional conv and masked as attention
        keys in every layer; their own output rows are unspecified (the splice never reads them)."""
        cfg, w = self.cfg, self.w
        B, N = wav.shape
        x2d, Tin, cin = wav.contiguous().view(B * N, 1), N, 1
        group_mode = cfg.get("hub_extractor_mode", "layer_norm") == "default"
        pre_ln = cfg.get("hub_layer_norm_first", True)
        pitched = None if group_mode else self._conv_stack_pitched(x2d, B, N)
        for i, (co, k, st) in enumerate(zip(cfg["hub_conv_dim"], cfg["hub_conv_kernel"], cfg["hub_conv_stride"])):
            if pitched is not None:     # the whole stack ran on per-clip pitches: one launch per layer (see _conv_stack_pitched)
                x2d, Tin, cin = pitched
                break
            if self._conv_by_view(i, cin, k):
                y = self._conv_view_gemm(x2d, B, Tin, cin, k, st, i, act=ACT_GELU if group_mode else ACT_NONE)
                x2d = y if group_mode else ops.layernorm(y, w[f"c{i}_lw"], w[f"c{i}_lb"], 1e-5, out=y, gelu=True)
                Tin, cin = (Tin - k) // st + 1, co
                continue
            cols, Tout = ops.conv1d_im2col(x2d, B, Tin, 0, cin, k, st, 0, Kp=w[f"c{i}"].shape[1])
            if not group_mode:      # "layer_norm" extractor: conv -> LayerNorm over channels -> GELU, every layer
                y = ops.gemm_nt(cols, w[f"c{i}"], bias=w[f"c{i}_b"])
                del cols
                x2d = ops.layernorm(y, w[f"c{i}_lw"], w[f"c{i}_lb"], 1e-5, out=y, gelu=True)
            elif i == 0:            # "default" extractor: GroupNorm (one group per channel, over TIME) after the first conv only;
                y = ops.gemm_nt(cols, w[f"c{i}"], bias=w[f"c{i}_b"], out_dtype=torch.float32)   # statistics from the fp32 product
                del cols
                x2d = ops.groupnorm_time_gelu(y, B, Tout, w[f"c{i}_lw"], w[f"c{i}_lb"], 1e-5)
                del y
            else:                   # ... the other layers are conv -> GELU (fused in the GEMM epilogue)
                x2d = ops.gemm_nt(cols, w[f"c{i}"], bias=w[f"c{i}_b"], act=ACT_GELU)
                del cols
            Tin, cin = Tout, co
        T, d, H, eps = Tin, cfg["hub_dim"], cfg["hub_heads"], cfg["hub_eps"]
        M = B * T
        h = ops.layernorm(x2d, w["fp_lw"], w["fp_lb"], eps)
        h = ops.gemm_nt(h, w["fp"], bias=w["fp_b"])
        key_mask = None
        if n_valid is not None:
            keep = self.valid_frames(N, n_valid)
            idx = torch.arange(M, dtype=torch.int32).view(B, T)
            km = torch.zeros((B, round_up(T, 64)), dtype=torch.uint8)
            for b_, kf in enumerate(keep):
                idx[b_, kf:] = -1
                km[b_, :kf] = 1
            h = ops.gather_rows(h, idx.view(-1).to(wav.device, non_blocking=True))   # padded frames -> zero rows
            key_mask = km.to(wav.device, non_blocking=True)
        G, kpos = cfg["hub_pos_groups"], cfg["hub_pos_k"]
        gch = d // G
        x = torch.empty((M, d), dtype=torch.bfloat16, device=wav.device)
        if POS_CONV_FUSED and ops.pos_conv_supported(gch, kpos):
            # ONE implicit-GEMM launch: the taps are row offsets into an LDS window, no im2col buffer (csrc/conv.hip)
            if "pos_tap" not in w:
                w["pos_tap"] = ops.pos_conv_pack(w["pos"], kpos)
            ops.pos_conv_fwd(h, w["pos_tap"], w["pos_b"], B, T, out=x)
        else:
            cols = torch.empty((M, self.pos_kp), dtype=torch.bfloat16, device=wav.device)
            for g in range(G):  # grouped conv: x[:, grp] = h[:, grp] + gelu(conv_g(h[:, grp]) + b_g)
                ops.conv1d_im2col(h, B, T, g * gch, gch, kpos, 1, kpos // 2, Kp=self.pos_kp, Tout_limit=T, out=cols)
                ops.gemm_nt(cols, w["pos"][g], out=x[:, g * gch:(g + 1) * gch], bias=w["pos_b"][g * gch:(g + 1) * gch],
                            act=ACT_GELU, residual=h[:, g * gch:(g + 1) * gch])
            del cols
        del h
        scale = 64 ** -0.5
        hbuf = torch.empty((M, d), dtype=torch.bfloat16, device=wav.device)
        qkv = torch.empty((M, 3 * d), dtype=torch.bfloat16, device=wav.device)
        obuf = torch.empty((M, d), dtype=torch.bfloat16, device=wav.device)
        fbuf = torch.empty((M, cfg["hub_ffn"]), dtype=torch.bfloat16, device=wav.device)
        if not pre_ln:      # post-LN encoders (Base): the encoder-level LayerNorm comes BEFORE the layers (WavLM.py:582-583)
            ops.layernorm(x, w["lnp_w"], w["lnp_b"], eps, out=x)
        for i in range(cfg["hub_layers"]):
            if pre_ln:
                ops.layernorm(x, w[f"{i}.ln1_w"], w[f"{i}.ln1_b"], eps, out=hbuf)
            attn_in = hbuf if pre_ln else x
            ops.gemm_nt(attn_in, w[f"{i}.qkv"], out=qkv, bias=w[f"{i}.qkv_b"])
            ops.attn_fwd(qkv[:, :d], qkv[:, d: 2 * d], qkv[:, 2 * d:], B, T, H, H, 64, False, scale, key_mask=key_mask, want_lse=False, out=obuf,
                         relpos=self._relpos(i, attn_in, B, T))
            if pre_ln:      # x += attn(LN1(x)); x += ffn(LN2(x))
                ops.gemm_nt(obuf, w[f"{i}.out"], out=x, bias=w[f"{i}.out_b"], residual=x)
                ops.layernorm(x, w[f"{i}.ln2_w"], w[f"{i}.ln2_b"], eps, out=hbuf)
                ops.gemm_nt(hbuf, w[f"{i}.fc1"], out=fbuf, bias=w[f"{i}.fc1_b"], act=ACT_GELU)
                ops.gemm_nt(fbuf, w[f"{i}.fc2"], out=x, bias=w[f"{i}.fc2_b"], residual=x)
            else:           # x = LN1(x + attn(x)); x = LN2(x + ffn(x))   (WavLM.py:716-739)
                ops.gemm_nt(obuf, w[f"{i}.out"], out=hbuf, bias=w[f"{i}.out_b"], residual=x)
                ops.layernorm(hbuf, w[f"{i}.ln1_w"], w[f"{i}.ln1_b"], eps, out=x)
                ops.gemm_nt(x, w[f"{i}.fc1"], out=fbuf, bias=w[f"{i}.fc1_b"], act=ACT_GELU)
                ops.gemm_nt(fbuf, w[f"{i}.fc2"], out=hbuf, bias=w[f"{i}.fc2_b"], residual=x)
                ops.layernorm(hbuf, w[f"{i}.ln2_w"], w[f"{i}.ln2_b"], eps, out=x)
        out = ops.layernorm(x, w["lnp_w"], w["lnp_b"], eps) if pre_ln else x
        return out.view(B, T, d)

    def _relpos(self, layer: int, attn_in: torch.Tensor, B: int, T: int):
        """additive attention bias of layer `layer` given its input (HuBERT: none)"""
        return None

    # ---- training forward / hand-written backward (trainable form) ----------------------------------------------------------
    def forward_train(self, wav: torch.Tensor, stash: dict, n_valid: Optional[List[int]] = None) -> torch.Tensor:
        """forward_wav that keeps what backward_hip() needs in stash["encoder"]: per conv layer its input and the conv output with the
        LayerNorm statistics (the normalised / GELU'd tensors are recomputed), the positional conv's pre-activation, and per
        transformer layer the same set as the Whisper encoder's training forward."""
        cfg, w = self.cfg, self.w
        B, N = wav.shape
        x2d, Tin, cin = wav.contiguous().view(B * N, 1), N, 1
        convs = []
        for i, (co, k, st) in enumerate(zip(cfg["hub_conv_dim"], cfg["hub_conv_kernel"], cfg["hub_conv_stride"])):
            by_view = self._conv_by_view(i, cin, k)
            Tout = (Tin - k) // st + 1
            cols = None if by_view else ops.conv1d_im2col(x2d, B, Tin, 0, cin, k, st, 0, Kp=w[f"c{i}"].shape[1])[0]
            if by_view and (not self._group_mode or i > 0):
                y = self._conv_view_gemm(x2d, B, Tin, cin, k, st, i)
                if not self._group_mode:
                    z, m, r = ops.layernorm(y, w[f"c{i}_lw"], w[f"c{i}_lb"], 1e-5, stats=True)
                    xo = ops.gelu_fwd(z)
                    del z
                    convs.append(dict(x_in=x2d, Tin=Tin, cin=cin, y=y, m=m, r=r))
                else:
                    xo = ops.gelu_fwd(y)
                    convs.append(dict(x_in=x2d, Tin=Tin, cin=cin, y=y))
            elif not self._group_mode:      # conv -> LayerNorm over channels -> GELU
                y = ops.gemm_nt(cols, w[f"c{i}"], bias=w[f"c{i}_b"])
                z, m, r = ops.layernorm(y, w[f"c{i}_lw"], w[f"c{i}_lb"], 1e-5, stats=True)
                xo = ops.gelu_fwd(z)
                del z
                convs.append(dict(x_in=x2d, Tin=Tin, cin=cin, y=y, m=m, r=r))
            elif i == 0:                  # conv -> GroupNorm over time (statistics from the fp32 product) -> GELU
                y = ops.gemm_nt(cols, w[f"c{i}"], bias=w[f"c{i}_b"], out_dtype=torch.float32)
                xo, gstats = ops.groupnorm_time_gelu(y, B, Tout, w[f"c{i}_lw"], w[f"c{i}_lb"], 1e-5, stats=True)
                convs.append(dict(x_in=x2d, Tin=Tin, cin=cin, y=y, gstats=gstats, Tout=Tout))
            else:                         # conv -> GELU: the pre-activation is kept for the backward
                y = ops.gemm_nt(cols, w[f"c{i}"], bias=w[f"c{i}_b"])
                xo = ops.gelu_fwd(y)
                convs.append(dict(x_in=x2d, Tin=Tin, cin=cin, y=y))
            del cols
            x2d, Tin, cin = xo, Tout, co
        T, d, H, eps = Tin, cfg["hub_dim"], cfg["hub_heads"], cfg["hub_eps"]
        M = B * T
        hN, mf, rf = ops.layernorm(x2d, w["fp_lw"], w["fp_lb"], eps, stats=True)
        h = ops.gemm_nt(hN, w["fp"], bias=w["fp_b"])
        key_mask = pad_idx = None
        if n_valid is not None:
            keep = self.valid_frames(N, n_valid)
            idx = torch.arange(M, dtype=torch.int32).view(B, T)
            km = torch.zeros((B, round_up(T, 64)), dtype=torch.uint8)
            for b_, kf in enumerate(keep):
                idx[b_, kf:] = -1
                km[b_, :kf] = 1
            pad_idx = idx.view(-1).to(wav.device, non_blocking=True)
            h = ops.gather_rows(h, pad_idx)                        # padded frames -> zero rows
            key_mask = km.to(wav.device, non_blocking=True)
        kpos = cfg["hub_pos_k"]
        RG = self._regularisers()                      # all-None keys in eval mode / frozen / p = 0: the deterministic graph
        k_in = RG.key(RG.p_in)
        if k_in is not None:                           # dropout_input (WavLM.py:353), before the padded frames are zeroed: same result
            h = ops.dropout(h, *k_in)
        pre = torch.empty((M, d), dtype=torch.bfloat16, device=wav.device)
        x = ops.pos_conv_fwd(h, w["pos_tap"], w["pos_b"], B, T, pre=pre)
        S = {"convs": convs, "x6": x2d, "mf": mf, "rf": rf, "hN": hN, "h": h, "pre": pre, "pad_idx": pad_idx, "key_mask": key_mask,
             "B": B, "T": T, "blocks": [], "k_in": k_in, "k_x": None}
        scale = 64 ** -0.5
        if not self._pre_ln:
            return self._forward_train_post_ln(x, S, stash, RG)
        S["k_x"] = RG.key(RG.p)                        # F.dropout after the positional conv (WavLM.py:584; layer_norm_first: no LayerNorm here)
        if S["k_x"] is not None:
            x = ops.dropout(x, *S["k_x"])
        for i in range(cfg["hub_layers"]):
            if not RG.keep_layer():                    # layerdrop: the layer is the identity this step
                S["blocks"].append(None)
                continue
            ka, k1, k2, k3 = RG.attn_key(RG.p_attn), RG.key(RG.p), RG.key(RG.p_act), RG.key(RG.p)
            hh, m1, r1 = ops.layernorm(x, w[f"{i}.ln1_w"], w[f"{i}.ln1_b"], eps, stats=True)
            qkv = ops.gemm_nt(hh, w[f"{i}.qkv"], bias=w[f"{i}.qkv_b"])
            # WavLM: (gate [B,H,Tp], bias table, T); HuBERT: None.  The bias is created by layer 0's attention and handed down
            # (WavLM.py:593-599): with layer 0 dropped by layerdrop the reference runs the other layers without it -- so does this
            rp = self._relpos(i, hh, B, T) if i == 0 or S["blocks"][0] is not None else None
            a, lse = ops.attn_fwd(qkv[:, :d], qkv[:, d: 2 * d], qkv[:, 2 * d:], B, T, H, H, 64, False, scale, key_mask=key_mask, want_lse=True,
                                  relpos=rp, drop=ka)
            if k1 is None:
                x1 = ops.gemm_nt(a, w[f"{i}.out"], bias=w[f"{i}.out_b"], residual=x)
            else:
                x1 = self._drop_add(ops.gemm_nt(a, w[f"{i}.out"], bias=w[f"{i}.out_b"]), x, k1)
            h2, m2, r2 = ops.layernorm(x1, w[f"{i}.ln2_w"], w[f"{i}.ln2_b"], eps, stats=True)
            z = ops.gemm_nt(h2, w[f"{i}.fc1"], bias=w[f"{i}.fc1_b"])
            fo = ops.gelu_fwd(z)
            if k2 is not None:
                fo = ops.dropout(fo, *k2)
            if k3 is None:
                x2 = ops.gemm_nt(fo, w[f"{i}.fc2"], bias=w[f"{i}.fc2_b"], residual=x1)
            else:
                x2 = self._drop_add(ops.gemm_nt(fo, w[f"{i}.fc2"], bias=w[f"{i}.fc2_b"]), x1, k3)
            del fo
            S["blocks"].append(dict(x=x, m1=m1, r1=r1, h=hh, qkv=qkv, a=a, lse=lse, x1=x1, m2=m2, r2=r2, h2=h2, z=z, rp=rp,
                                    ka=ka, k1=k1, k2=k2, k3=k3))
            x = x2
        out, mo, ro = ops.layernorm(x, w["lnp_w"], w["lnp_b"], eps, stats=True)
        S.update(x_last=x, mo=mo, ro=ro)
        stash["encoder"] = S
        return out.view(B, T, d)

    def _forward_train_post_ln(self, x: torch.Tensor, S: dict, stash: dict, RG: SimpleNamespace) -> torch.Tensor:
        """the layers of the post-LN geometries (Base): encoder LayerNorm first, then x = LN1(x + attn(x)); x = LN2(x + ffn(x)) per layer
        (WavLM.py:582-583 / :716-739, modeling_hubert.py:395-420, 470-520), no LayerNorm after them"""
        cfg, w = self.cfg, self.w
        B, T, d, H, eps = S["B"], S["T"], cfg["hub_dim"], cfg["hub_heads"], cfg["hub_eps"]
        scale = 64 ** -0.5
        x_pre = x
        x, mo, ro = ops.layernorm(x_pre, w["lnp_w"], w["lnp_b"], eps, stats=True)
        S.update(x_pre=x_pre, mo=mo, ro=ro)
        S["k_x"] = RG.key(RG.p)                        # F.dropout after the encoder LayerNorm (WavLM.py:582-584)
        if S["k_x"] is not None:
            x = ops.dropout(x, *S["k_x"])
        for i in range(cfg["hub_layers"]):
            if not RG.keep_layer():
                S["blocks"].append(None)
                continue
            ka, k1, k2, k3 = RG.attn_key(RG.p_attn), RG.key(RG.p), RG.key(RG.p_act), RG.key(RG.p)
            qkv = ops.gemm_nt(x, w[f"{i}.qkv"], bias=w[f"{i}.qkv_b"])
            rp = self._relpos(i, x, B, T) if i == 0 or S["blocks"][0] is not None else None
            a, lse = ops.attn_fwd(qkv[:, :d], qkv[:, d: 2 * d], qkv[:, 2 * d:], B, T, H, H, 64, False, scale, key_mask=S["key_mask"], want_lse=True,
                                  relpos=rp, drop=ka)
            if k1 is None:
                s1 = ops.gemm_nt(a, w[f"{i}.out"], bias=w[f"{i}.out_b"], residual=x)
            else:
                s1 = self._drop_add(ops.gemm_nt(a, w[f"{i}.out"], bias=w[f"{i}.out_b"]), x, k1)
            x1, m1, r1 = ops.layernorm(s1, w[f"{i}.ln1_w"], w[f"{i}.ln1_b"], eps, stats=True)
            z = ops.gemm_nt(x1, w[f"{i}.fc1"], bias=w[f"{i}.fc1_b"])
            fo = ops.gelu_fwd(z)
            if k2 is not None:
                fo = ops.dropout(fo, *k2)
            if k3 is None:
                s2 = ops.gemm_nt(fo, w[f"{i}.fc2"], bias=w[f"{i}.fc2_b"], residual=x1)
            else:
                s2 = self._drop_add(ops.gemm_nt(fo, w[f"{i}.fc2"], bias=w[f"{i}.fc2_b"]), x1, k3)
            del fo
            x2, m2, r2 = ops.layernorm(s2, w[f"{i}.ln2_w"], w[f"{i}.ln2_b"], eps, stats=True)
            S["blocks"].append(dict(h=x, qkv=qkv, a=a, lse=lse, s1=s1, m1=m1, r1=r1, x1=x1, z=z, s2=s2, m2=m2, r2=r2, rp=rp,
                                    ka=ka, k1=k1, k2=k2, k3=k3))
            x = x2
        stash["encoder"] = S
        return x.view(B, T, d)

    def _lin_grads(self, dy: torch.Tensor, x: torch.Tensor, w_name: str, acc: bool, N: Optional[int] = None, K: Optional[int] = None,
                   bias: Tuple = ()):
        """dW (+)= dy^T x into the flat gradient buffer at parameter `w_name` (N x K may span the fused q|k|v block); bias = ((param
        name, first column of dy, width), ...) column sums"""
        st = self.store
        Mp = round_up(dy.shape[0], 64)
        off, _, shape = st.offsets[w_name]
        N, K = N or shape[0], K or shape[1]
        ops.gemm_nt(ops.transpose(dy, Rp=Mp), ops.transpose(x, Rp=Mp), out=st.grad[off: off + N * K].view(N, K), accumulate=acc)
        for name, c0, n in bias:
            ops.colsum(dy[:, c0: c0 + n], st.grad_view(name), accumulate=acc)

    def backward_hip(self, dout: torch.Tensor, stash: dict, acc: bool):
        """dout [B*T, d] bf16 = dL/d(encoder output); deposits every encoder gradient into the flat grad buffer (hand-written adjoint of
        the graph of forward_train: HF HubertModel / fairseq HubertModel.extract_features as called at slam_model.py:335-341, WavLM
        extract_features :333-334)"""
        cfg, w, st, N = self.cfg, self.w, self.store, self._nm()
        S = stash.pop("encoder")
        B, T, key_mask = S["B"], S["T"], S["key_mask"]
        d, H = cfg["hub_dim"], cfg["hub_heads"]
        scale = 64 ** -0.5
        gv = st.grad_view
        rp_state = self._relpos_backward_begin(S)
        if self._pre_ln:
            dx = ops.layernorm_bwd(S["x_last"], S["mo"], S["ro"], w["lnp_w"], dout, dgamma=gv(N.enc_ln + ".weight"),
                                   dbeta=gv(N.enc_ln + ".bias"), accumulate=acc)
        else:
            dx = self._backward_layers_post_ln(dout, S, rp_state, acc)
        for i in (reversed(range(cfg["hub_layers"])) if self._pre_ln else ()):
            R = S["blocks"][i]
            if R is None:                              # layerdrop: identity this step -- its parameters get no gradient (zero-filled below)
                self._zero_layer_grads(i, acc)
                continue
            fo = ops.gelu_fwd(R["z"])
            if R["k2"] is not None:
                fo = ops.dropout(fo, *R["k2"])
            dt = dx if R["k3"] is None else ops.dropout(dx, *R["k3"])         # dL/d(fc2 output): the residual branch keeps the undropped dx
            self._lin_grads(dt, fo, N.fc2(i) + ".weight", acc, bias=((N.fc2(i) + ".bias", 0, d),))
            dfo = ops.gemm_nt(dt, w[f"{i}.fc2T"])
            if R["k2"] is not None:
                dfo = ops.dropout(dfo, *R["k2"])
            dz = ops.gelu_bwd(R["z"], dfo)
            del fo, dfo, dt
            self._lin_grads(dz, R["h2"], N.fc1(i) + ".weight", acc, bias=((N.fc1(i) + ".bias", 0, cfg["hub_ffn"]),))
            dh2 = ops.gemm_nt(dz, w[f"{i}.fc1T"])
            del dz
            dx1 = ops.layernorm_bwd(R["x1"], R["m2"], R["r2"], w[f"{i}.ln2_w"], dh2, dgamma=gv(N.ln2(i) + ".weight"),
                                    dbeta=gv(N.ln2(i) + ".bias"), accumulate=acc)
            ops.add_(dx1, dx)
            dt = dx1 if R["k1"] is None else ops.dropout(dx1, *R["k1"])
            self._lin_grads(dt, R["a"], N.out(i) + ".weight", acc, bias=((N.out(i) + ".bias", 0, d),))
            da = ops.gemm_nt(dt, w[f"{i}.outT"])
            del dt
            qkv = R["qkv"]
            dqkv = torch.empty_like(qkv)
            rp_b = self._relpos_backward_args(R, rp_state) if R["rp"] is not None else None   # WavLM: (gate, table, T, d_gate OUT, d_table ACCUMULATED)
            if R["rp"] is None:
                self._zero_gate_grads(i, acc)
            ops.attn_bwd(qkv[:, :d], qkv[:, d: 2 * d], qkv[:, 2 * d:], R["a"], da, R["lse"], dqkv[:, :d],
                         dqkv[:, d: 2 * d], dqkv[:, 2 * d:], B, T, H, H, 64, False, scale, key_mask=key_mask, relpos=rp_b, drop=R["ka"])
            del da
            self._lin_grads(dqkv, R["h"], N.q(i) + ".weight", acc, N=3 * d, K=d,
                            bias=((N.q(i) + ".bias", 0, d), (N.k(i) + ".bias", d, d), (N.v(i) + ".bias", 2 * d, d)))
            dh = ops.gemm_nt(dqkv, w[f"{i}.qkvT"])
            del dqkv
            if rp_b is not None:    # the gate is a function of the attention input too
                ops.add_(dh, self._gate_backward(i, R, rp_b[3], acc))
            dx = ops.layernorm_bwd(R["x"], R["m1"], R["r1"], w[f"{i}.ln1_w"], dh, dgamma=gv(N.ln1(i) + ".weight"),
                                   dbeta=gv(N.ln1(i) + ".bias"), accumulate=acc)
            ops.add_(dx, dx1)
            S["blocks"][i] = None
        self._relpos_backward_end(rp_state, acc)
        # ---- positional conv: x0 = h + gelu(conv(h) + b) ----
        if self._pre_ln and S["k_x"] is not None:
            dx = ops.dropout(dx, *S["k_x"])
        G, kpos = cfg["hub_pos_groups"], cfg["hub_pos_k"]
        gch = d // G
        M = B * T
        Mp = round_up(M, 64)
        dpre = ops.gelu_bwd(S["pre"], dx)
        dw = torch.empty((d, gch, kpos), dtype=torch.float32, device=dx.device)     # reference layout [d (co), gch (ci), kpos]
        cols = torch.empty((M, self.pos_kp), dtype=torch.bfloat16, device=dx.device)
        for g in range(G):   # dW_g[co, j*gch + ci] = sum_t dpre[t, g*gch + co] * h[t + j - kpos/2, g*gch + ci]
            ops.conv1d_im2col(S["h"], B, T, g * gch, gch, kpos, 1, kpos // 2, Kp=self.pos_kp, Tout_limit=T, out=cols)
            gwg = ops.gemm_nt(ops.transpose(dpre[:, g * gch:(g + 1) * gch], Rp=Mp), ops.transpose(cols, Rp=Mp), out_dtype=torch.float32)
            dw[g * gch:(g + 1) * gch].copy_(gwg[:, : kpos * gch].reshape(gch, kpos, gch).permute(0, 2, 1))
        del cols
        self._deposit_pos_weight_grad(dw, acc)
        ops.colsum(dpre, gv(N.pos_b), accumulate=acc)
        dh = ops.pos_conv_fwd(dpre, w["pos_adj"], None, B, T, residual=dx, pad=kpos - 1 - kpos // 2, act=False)
        del dpre, dx
        if S["pad_idx"] is not None:
            dh = ops.gather_rows(dh, S["pad_idx"])      # the padded frames were zero-filled in the forward: no gradient through them
        if S["k_in"] is not None:
            dh = ops.dropout(dh, *S["k_in"])
        # ---- feature projection: h = Linear(LayerNorm(x6)) ----
        self._lin_grads(dh, S["hN"], N.fp + ".weight", acc, bias=((N.fp + ".bias", 0, d),))
        dhN = ops.gemm_nt(dh, w["fpT"])
        del dh
        dxo = ops.layernorm_bwd(S["x6"], S["mf"], S["rf"], w["fp_lw"], dhN, dgamma=gv(N.fp_ln + ".weight"), dbeta=gv(N.fp_ln + ".bias"),
                                accumulate=acc)
        del dhN
        # ---- conv feature extractor, last layer first: x_i = gelu(LayerNorm(conv_i(x_{i-1}))) ----
        for i in reversed(range(len(cfg["hub_conv_dim"]))):
            C_ = S["convs"][i]
            co, k, sd = cfg["hub_conv_dim"][i], cfg["hub_conv_kernel"][i], cfg["hub_conv_stride"][i]
            if not self._group_mode:
                z = ops.layernorm(C_["y"], w[f"c{i}_lw"], w[f"c{i}_lb"], 1e-5)
                dz = ops.gelu_bwd(z, dxo)
                del z, dxo
                dy = ops.layernorm_bwd(C_["y"], C_["m"], C_["r"], w[f"c{i}_lw"], dz, dgamma=gv(N.conv_ln(i) + ".weight"),
                                       dbeta=gv(N.conv_ln(i) + ".bias"), accumulate=acc)
                del dz
            elif i == 0:
                dy = ops.groupnorm_time_gelu_bwd(C_["y"], C_["gstats"], w[f"c{i}_lw"], w[f"c{i}_lb"], dxo, B, C_["Tout"],
                                                 gv(N.conv_ln(i) + ".weight"), gv(N.conv_ln(i) + ".bias"), accumulate=acc)
                del dxo
            else:
                dy = ops.gelu_bwd(C_["y"], dxo)
                del dxo
            cin, Tin = C_["cin"], C_["Tin"]
            cols, Tout = ops.conv1d_im2col(C_["x_in"], B, Tin, 0, cin, k, sd, 0, Kp=w[f"c{i}"].shape[1])
            Mi = dy.shape[0]
            Mip = round_up(Mi, 64)
            gwc = ops.gemm_nt(ops.transpose(dy, Rp=Mip), ops.transpose(cols, Rp=Mip), out_dtype=torch.float32)   # [co, kp], tap-major
            del cols
            gwc = gwc[:, : k * cin].reshape(co, k, cin).permute(0, 2, 1)
            gv(N.conv_w(i)).add_(gwc) if acc else gv(N.conv_w(i)).copy_(gwc)
            if N.conv_b is not None:
                ops.colsum(dy, gv(N.conv_b(i)), accumulate=acc)
            if i > 0:
                dxo = ops.conv1d_col2im(ops.gemm_nt(dy, w[f"c{i}T"]), B, Tin, cin, k, sd)
            del dy
            S["convs"][i] = None

    def _backward_layers_post_ln(self, dout: torch.Tensor, S: dict, rp_state, acc: bool) -> torch.Tensor:
        """adjoint of _forward_train_post_ln: returns dL/d(positional conv output)"""
        cfg, w, st, N = self.cfg, self.w, self.store, self._nm()
        B, T, key_mask = S["B"], S["T"], S["key_mask"]
        d, H = cfg["hub_dim"], cfg["hub_heads"]
        scale = 64 ** -0.5
        gv = st.grad_view
        dx = dout
        for i in reversed(range(cfg["hub_layers"])):
            R = S["blocks"][i]
            if R is None:
                self._zero_layer_grads(i, acc)
                continue
            ds2 = ops.layernorm_bwd(R["s2"], R["m2"], R["r2"], w[f"{i}.ln2_w"], dx, dgamma=gv(N.ln2(i) + ".weight"),
                                    dbeta=gv(N.ln2(i) + ".bias"), accumulate=acc)
            fo = ops.gelu_fwd(R["z"])
            if R["k2"] is not None:
                fo = ops.dropout(fo, *R["k2"])
            dt = ds2 if R["k3"] is None else ops.dropout(ds2, *R["k3"])
            self._lin_grads(dt, fo, N.fc2(i) + ".weight", acc, bias=((N.fc2(i) + ".bias", 0, d),))
            dfo = ops.gemm_nt(dt, w[f"{i}.fc2T"])
            if R["k2"] is not None:
                dfo = ops.dropout(dfo, *R["k2"])
            dz = ops.gelu_bwd(R["z"], dfo)
            del fo, dfo, dt
            self._lin_grads(dz, R["x1"], N.fc1(i) + ".weight", acc, bias=((N.fc1(i) + ".bias", 0, cfg["hub_ffn"]),))
            dx1 = ops.gemm_nt(dz, w[f"{i}.fc1T"])
            del dz
            ops.add_(dx1, ds2)
            del ds2
            ds1 = ops.layernorm_bwd(R["s1"], R["m1"], R["r1"], w[f"{i}.ln1_w"], dx1, dgamma=gv(N.ln1(i) + ".weight"),
                                    dbeta=gv(N.ln1(i) + ".bias"), accumulate=acc)
            del dx1
            dt = ds1 if R["k1"] is None else ops.dropout(ds1, *R["k1"])
            self._lin_grads(dt, R["a"], N.out(i) + ".weight", acc, bias=((N.out(i) + ".bias", 0, d),))
            da = ops.gemm_nt(dt, w[f"{i}.outT"])
            del dt
            qkv = R["qkv"]
            dqkv = torch.empty_like(qkv)
            rp_b = self._relpos_backward_args(R, rp_state) if R["rp"] is not None else None
            if R["rp"] is None:
                self._zero_gate_grads(i, acc)
            ops.attn_bwd(qkv[:, :d], qkv[:, d: 2 * d], qkv[:, 2 * d:], R["a"], da, R["lse"], dqkv[:, :d],
                         dqkv[:, d: 2 * d], dqkv[:, 2 * d:], B, T, H, H, 64, False, scale, key_mask=key_mask, relpos=rp_b, drop=R["ka"])
            del da
            self._lin_grads(dqkv, R["h"], N.q(i) + ".weight", acc, N=3 * d, K=d,
                            bias=((N.q(i) + ".bias", 0, d), (N.k(i) + ".bias", d, d), (N.v(i) + ".bias", 2 * d, d)))
            dx = ops.gemm_nt(dqkv, w[f"{i}.qkvT"])
            del dqkv
            ops.add_(dx, ds1)
            if rp_b is not None:
                ops.add_(dx, self._gate_backward(i, R, rp_b[3], acc))
            S["blocks"][i] = None
        if S["k_x"] is not None:
            dx = ops.dropout(dx, *S["k_x"])
        return ops.layernorm_bwd(S["x_pre"], S["mo"], S["ro"], w["lnp_w"], dx, dgamma=gv(N.enc_ln + ".weight"), dbeta=gv(N.enc_ln + ".bias"),
                                 accumulate=acc)

    # hooks of the relative-position-bias backward (WavLM overrides; HuBERT has no bias)
    def _relpos_backward_begin(self, S: dict):
        return None

    def _relpos_backward_args(self, R: dict, state):
        return None

    def _gate_backward(self, i: int, R: dict, d_gate: torch.Tensor, acc: bool):
        raise NotImplementedError

    def _relpos_backward_end(self, state, acc: bool):
        pass

    def _zero_gate_grads(self, i: int, acc: bool):
        pass

    def _zero_layer_grads(self, i: int, acc: bool):
        """a layer skipped by layerdrop contributes no gradient: with acc=False its slots of the flat buffer still hold the previous
        step's values and are cleared (autograd would leave .grad = None)"""
        if acc:
            return
        pre = self._nm().fc1(i).split(f"layers.{i}.")[0] + f"layers.{i}."
        for name in self.store.offsets:
            if name.startswith(pre):
                self.store.grad_view(name).zero_()

    def _deposit_pos_weight_grad(self, dw: torch.Tensor, acc: bool):
        st, N = self.store, self._nm()
        if N.pos_w is None:     # weight_norm(dim=2) kept as (g, v): chain rule of w = g v / ||v|| (fairseq / reference WavLM names)
            ops.weight_norm_bwd(dw, st.master_view(N.pos_v), st.master_view(N.pos_g), st.grad_view(N.pos_g), st.grad_view(N.pos_v), accumulate=acc)
            return
        g = st.grad_view(N.pos_w)
        g.add_(dw) if acc else g.copy_(dw)

    def forward(self, source=None, padding_mask=None, **kw):
        return {"encoder_out": self.forward_wav(source).transpose(0, 1), "padding_mask": None}


class HipWavLMEncoder(HipHubertEncoder):
    """Frozen WavLM encoder (src/slam_llm/models/wavlm/WavLM.py:220-376 as called through models/encoder.py:109-127 from
    models/slam_model.py:333-334).  Base / Base+ (`hub_extractor_mode="default"`, `hub_layer_norm_first=False`): GroupNorm over
    time after the first conv only (slam_groupnorm_time_gelu), conv -> GELU for the rest, post-LN layers with the encoder-level
    LayerNorm in front of them.  WavLM-Large's configuration: "layer_norm" conv feature extractor without conv bias, feature
    LayerNorm + projection, weight-normed grouped positional conv, layer_norm_first transformer -- i.e. HuBERT-large's graph (the
    parent class) -- plus the gated relative position bias in every attention (modules.py:504-533): layer 0's bucketed
    `relative_attention_bias` is laid out once per sequence length as a per-head table over the relative distance k - q, each
    layer's gate comes from slam_wavlm_gate on that layer's attention input, and the attention kernel adds gate[q] * table[k - q]
    to the scores.  Weights are read under the reference module's own state-dict names (`encoder.model.*`)."""

    def __init__(self, cfg: dict, device, store: Optional["TrainableStore"] = None, prefix="encoder.model."):
        super().__init__(cfg, device, store, prefix)
        self._tables, self._buckets = {}, {}

    # ---- trainable form: the reference module's own parameter names (WavLM.py:220-330, modules.py:330-420) -------------------
    def _nm(self) -> SimpleNamespace:
        N = self._nm_fairseq()      # WavLM.py vendors fairseq's module tree: same names, no conv bias, plus the gate / bucket table / mask_emb
        p = self.prefix
        lyr = lambda i: f"{p}encoder.layers.{i}."                # noqa: E731
        N.conv_b = None
        N.grep_w = lambda i: lyr(i) + "self_attn.grep_linear.weight"
        N.grep_b = lambda i: lyr(i) + "self_attn.grep_linear.bias"
        N.grep_a = lambda i: lyr(i) + "self_attn.grep_a"
        N.rel_bias = lyr(0) + "self_attn.relative_attention_bias.weight"
        N.mask_emb = p + "mask_emb"
        return N

    def _reserve_extra_layer(self, i: int):
        cfg, r, N = self.cfg, self.store.reserve, self._nm()
        H = cfg["hub_heads"]
        r(N.grep_w(i), (8, 64)); r(N.grep_b(i), (8,)); r(N.grep_a(i), (1, H, 1, 1))
        if i == 0:      # only layer 0 owns the bucket embedding (has_relative_attention_bias, WavLM.py:606-620); the others reuse its table
            r(N.rel_bias, (cfg["wavlm_buckets"], H))

    def _reserve_extra_top(self):
        # mask_emb (WavLM.py:262-264) is a parameter of the module, so the optimizer sees it; extract_features(mask=False) never reads
        # it: its gradient stays zero, like autograd's None
        self.store.reserve(self._nm().mask_emb, (self.cfg["hub_dim"],))

    def _load_trainable(self, W: Dict[str, torch.Tensor], prefix: str):
        N = self._nm()
        if prefix + N.pos_g[len(self.prefix):] not in W:     # a checkpoint saved under the parametrizations API's names
            base = prefix + N.pos_g[len(self.prefix): -len("weight_g")]
            W = dict(W)
            W[base + "weight_g"], W[base + "weight_v"] = W[base + "parametrizations.weight.original0"], W[base + "parametrizations.weight.original1"]
        if prefix + N.mask_emb[len(self.prefix):] not in W:
            W = dict(W)
            W[prefix + N.mask_emb[len(self.prefix):]] = torch.zeros(self.cfg["hub_dim"])
        return super()._load_trainable(W, prefix)

    def _refresh_extra(self):
        st, w, N, H = self.store, self.w, self._nm(), self.cfg["hub_heads"]
        for i in range(self.cfg["hub_layers"]):
            w[f"{i}.gw"], w[f"{i}.gb"] = st.master_view(N.grep_w(i)), st.master_view(N.grep_b(i))
            w[f"{i}.ga"] = st.master_view(N.grep_a(i)).view(H)
        w["rel_bias"] = st.master_view(N.rel_bias)
        self._tables = {}      # the per-length bias table is a function of the (now moving) bucket embedding

    def _relpos_backward_begin(self, S: dict):
        tab = self._tables.get(S["T"])
        if tab is None:         # layer 0 dropped on the first step at this length: no layer used the bias
            tab = self._table(S["T"])
        return dict(tab=tab, d_tab=torch.zeros_like(tab), T=S["T"])

    def _zero_gate_grads(self, i: int, acc: bool):
        if not acc:
            N = self._nm()
            for name in (N.grep_w(i), N.grep_b(i), N.grep_a(i)):
                self.store.grad_view(name).zero_()

    def _relpos_backward_args(self, R: dict, state):
        gate = R["rp"][0]
        return (gate, state["tab"], state["T"], torch.empty_like(gate), state["d_tab"])

    def _gate_backward(self, i: int, R: dict, d_gate: torch.Tensor, acc: bool):
        """adjoint of slam_wavlm_gate (modules.py:504-533): gate = ga * (sigmoid(sum4 a) * (sigmoid(sum4 b) * grep_a - 1) + 2) with
        (a | b) = grep_linear(x viewed per head); deposits grep_linear / grep_a gradients, returns dL/d(attention input)"""
        st, w, N, H = self.store, self.w, self._nm(), self.cfg["hub_heads"]
        B, T = R["rp"][0].shape[0], R["rp"][2]
        x = R["h"]
        dv8, da_term, dxg = ops.wavlm_gate_bwd(x, w[f"{i}.gw"], w[f"{i}.gb"], w[f"{i}.ga"], d_gate, B, T, H)
        ops.skinny_gram(dv8, x.view(B * T * H, 64), st.grad_view(N.grep_w(i)), 64, 1, accumulate=acc)
        ops.colsum(dv8, st.grad_view(N.grep_b(i)), accumulate=acc)
        tmp = torch.empty(da_term.shape[1], dtype=torch.float32, device=x.device)
        ops.colsum(da_term, tmp)
        ga = st.grad_view(N.grep_a(i)).view(H)
        ga.add_(tmp[:H]) if acc else ga.copy_(tmp[:H])
        return dxg

    def _relpos_backward_end(self, state, acc: bool):
        ops.relpos_bucket_grad(state["d_tab"], self._buckets[state["T"]], self.cfg["wavlm_buckets"],
                               self.store.grad_view(self._nm().rel_bias), accumulate=acc)

    def load(self, W: Dict[str, torch.Tensor], prefix="encoder.model."):
        if self.trainable:
            return self._load_trainable(W, prefix)
        cfg, dev, w = self.cfg, self.device_, self.w
        bf = lambda t: t.to(device=dev, dtype=torch.bfloat16).contiguous()  # noqa: E731
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        cin = 1
        for i, (co, k) in enumerate(zip(cfg["hub_conv_dim"], cfg["hub_conv_kernel"])):
            p = f"{prefix}feature_extractor.conv_layers.{i}."
            kp = round_up(k * cin, 64)
            wc = torch.zeros((co, kp), dtype=torch.bfloat16, device=dev)
            wc[:, : k * cin] = bf(W[p + "0.weight"].permute(0, 2, 1).reshape(co, k * cin))
            cb = W.get(p + "0.bias")                                  # conv_bias=False in the released WavLM configurations
            w[f"c{i}"], w[f"c{i}_b"] = wc, (f32(cb) if cb is not None else torch.zeros(co, dtype=torch.float32, device=dev))
            if cfg.get("hub_extractor_mode", "layer_norm") == "default":    # Base: GroupNorm after the first conv only
                if i == 0:
                    w[f"c{i}_lw"], w[f"c{i}_lb"] = f32(W[p + "2.weight"]), f32(W[p + "2.bias"])
            else:
                w[f"c{i}_lw"], w[f"c{i}_lb"] = f32(W[p + "2.1.weight"]), f32(W[p + "2.1.bias"])
            cin = co
        d, H = cfg["hub_dim"], cfg["hub_heads"]
        assert d % 64 == 0 and d // H == 64 and cin % 64 == 0
        w["fp_lw"], w["fp_lb"] = f32(W[prefix + "layer_norm.weight"]), f32(W[prefix + "layer_norm.bias"])
        w["fp"], w["fp_b"] = bf(W[prefix + "post_extract_proj.weight"]), f32(W[prefix + "post_extract_proj.bias"])
        p = prefix + "encoder."
        if p + "pos_conv.0.weight_g" in W:      # nn.utils.weight_norm(dim=2): w = g * v / ||v|| over dims (0, 1)
            g_, v_ = W[p + "pos_conv.0.weight_g"].float(), W[p + "pos_conv.0.weight_v"].float()
        else:                                   # the parametrizations API's names for the same two tensors
            g_, v_ = W[p + "pos_conv.0.parametrizations.weight.original0"].float(), W[p + "pos_conv.0.parametrizations.weight.original1"].float()
        pw = g_ * v_ / v_.norm(dim=(0, 1), keepdim=True)
        G, kpos = cfg["hub_pos_groups"], cfg["hub_pos_k"]
        gch = d // G
        self.pos_kp = round_up(kpos * gch, 64)
        pg = torch.zeros((G, gch, self.pos_kp), dtype=torch.bfloat16, device=dev)
        for g in range(G):
            pg[g, :, : kpos * gch] = bf(pw[g * gch:(g + 1) * gch].permute(0, 2, 1).reshape(gch, kpos * gch))
        w["pos"], w["pos_b"] = pg, f32(W[p + "pos_conv.0.bias"])
        for i in range(cfg["hub_layers"]):
            q = f"{p}layers.{i}."
            a = q + "self_attn."
            w[f"{i}.qkv"] = bf(torch.cat([W[a + "q_proj.weight"], W[a + "k_proj.weight"], W[a + "v_proj.weight"]], 0))
            w[f"{i}.qkv_b"] = f32(torch.cat([W[a + "q_proj.bias"], W[a + "k_proj.bias"], W[a + "v_proj.bias"]], 0))
            w[f"{i}.out"], w[f"{i}.out_b"] = bf(W[a + "out_proj.weight"]), f32(W[a + "out_proj.bias"])
            w[f"{i}.ln1_w"], w[f"{i}.ln1_b"] = f32(W[q + "self_attn_layer_norm.weight"]), f32(W[q + "self_attn_layer_norm.bias"])
            w[f"{i}.fc1"], w[f"{i}.fc1_b"] = bf(W[q + "fc1.weight"]), f32(W[q + "fc1.bias"])
            w[f"{i}.fc2"], w[f"{i}.fc2_b"] = bf(W[q + "fc2.weight"]), f32(W[q + "fc2.bias"])
            w[f"{i}.ln2_w"], w[f"{i}.ln2_b"] = f32(W[q + "final_layer_norm.weight"]), f32(W[q + "final_layer_norm.bias"])
            w[f"{i}.gw"], w[f"{i}.gb"] = f32(W[a + "grep_linear.weight"]), f32(W[a + "grep_linear.bias"])
            w[f"{i}.ga"] = f32(W[a + "grep_a"].reshape(H))
        w["rel_bias"] = f32(W[p + "layers.0.self_attn.relative_attention_bias.weight"])      # [buckets, H]
        w["lnp_w"], w["lnp_b"] = f32(W[p + "layer_norm.weight"]), f32(W[p + "layer_norm.bias"])
        self._tables = {}
        return self

    def init_random(self, seed: int = 42):
        super().init_random(seed)
        cfg, dev, w = self.cfg, self.device_, self.w
        g = torch.Generator(device=dev).manual_seed(seed + 17)
        H = cfg["hub_heads"]
        for i in range(cfg["hub_layers"]):
            w[f"{i}.gw"] = torch.randn(8, 64, generator=g, device=dev) * 0.1
            w[f"{i}.gb"] = torch.randn(8, generator=g, device=dev) * 0.1
            w[f"{i}.ga"] = torch.ones(H, device=dev)
        for i in range(len(cfg["hub_conv_dim"])):
            w[f"c{i}_b"] = torch.zeros_like(w[f"c{i}_b"])
        w["rel_bias"] = torch.randn(cfg["wavlm_buckets"], H, generator=g, device=dev)
        self._tables = {}
        return self

    def _table(self, T: int) -> torch.Tensor:
        tab = self._tables.get(T)
        if tab is None:
            from .host_tables import wavlm_relative_buckets
            buckets = wavlm_relative_buckets(T, self.cfg["wavlm_buckets"], self.cfg["wavlm_max_distance"]).to(self.device_)
            tab = ops.relpos_table(self.w["rel_bias"].index_select(0, buckets).t().contiguous())      # [H, 2T-1] (+ slack)
            self._tables = {T: tab}
            self._buckets = {T: buckets.to(torch.int32)}
        return tab

    def _relpos(self, layer: int, attn_in: torch.Tensor, B: int, T: int):
        w, H = self.w, self.cfg["hub_heads"]
        tab = self._table(T)
        gate = ops.wavlm_gate(attn_in, w[f"{layer}.gw"], w[f"{layer}.gb"], w[f"{layer}.ga"], B, T, H)
        return (gate, tab, T)

# ======================================================================================== projector
class HipProjectorConcat(nn.Module):
    """EncoderProjectorConcat (src/slam_llm/models/projector.py:5-27): k-frame stack, Linear-ReLU-Linear."""

    def __init__(self, cfg: dict, store: TrainableStore, prefix="encoder_projector."):
        super().__init__()
        self.k, self.d, self.hid, self.dl = cfg["ds_rate"], cfg["enc_dim"], cfg["proj_hidden"], cfg["llm_dim"]
        self.store, self.prefix = store, prefix
        assert (self.k * self.d) % 64 == 0 and self.hid % 64 == 0 and self.dl % 64 == 0
        store.reserve(prefix + "linear1.weight", (self.hid, self.k * self.d))
        store.reserve(prefix + "linear1.bias", (self.hid,))
        store.reserve(prefix + "linear2.weight", (self.dl, self.hid))
        store.reserve(prefix + "linear2.bias", (self.dl,))
        self.linear1 = nn.Module()
        self.linear2 = nn.Module()
        self.need_dx = False   # True when the encoder is trainable: backward_hip then returns dL/d(encoder output)

    def bind(self):
        s, p = self.store, self.prefix
        self.linear1.weight, self.linear1.bias = s.params[p + "linear1.weight"], s.params[p + "linear1.bias"]
        self.linear2.weight, self.linear2.bias = s.params[p + "linear2.weight"], s.params[p + "linear2.bias"]

    def refresh(self):
        s, p = self.store, self.prefix
        # linear1's transpose only serves dL/d(encoder output): built when the encoder is trainable
        # (in place after the first call: see ops.transpose_into)
        self.w2T = ops.transpose_into(getattr(self, "w2T", None), s.bf16_view(p + "linear2.weight"), self.dl)  # [hid, dl]
        self.w1T = ops.transpose_into(getattr(self, "w1T", None), s.bf16_view(p + "linear1.weight"), self.hid) if self.need_dx else None  # [k*d, hid]

    def forward_hip(self, enc: torch.Tensor, stash: Optional[dict]):
        """enc [B, T2, d] bf16 -> [B, Ta, dl] bf16"""
        B, T2, d = enc.shape
        Ta = T2 // self.k
        xp = enc[:, : Ta * self.k, :]
        if T2 % self.k:
            xp = xp.contiguous()
        if stash is not None:
            stash["proj_shape"] = (B, T2)
        return self.forward_rows(xp.reshape(B * Ta, self.k * d), stash).view(B, Ta, self.dl)

    def forward_rows(self, xp: torch.Tensor, stash: Optional[dict]):
        """xp [rows, k*d] bf16: already k-frame stacked rows (any batch layout, e.g. the ragged encoder's windows) -> [rows, dl]"""
        s, p = self.store, self.prefix
        h = ops.gemm_nt(xp, s.bf16_view(p + "linear1.weight"), bias=s.master_view(p + "linear1.bias"), act=ACT_RELU)
        y = ops.gemm_nt(h, s.bf16_view(p + "linear2.weight"), bias=s.master_view(p + "linear2.bias"))
        if stash is not None:
            stash["proj"] = (xp, h)
        return y

    def _unstack(self, dxp: torch.Tensor, stash: dict):
        """dL/d(stacked rows) [B*Ta, k*d] -> dL/d(encoder output) [B*T2, d] (frames past Ta*k were dropped: zero gradient)"""
        B, T2 = stash.pop("proj_shape")
        Ta = T2 // self.k
        if T2 == Ta * self.k:
            return dxp.view(B * T2, self.d)
        out = torch.zeros((B, T2, self.d), dtype=dxp.dtype, device=dxp.device)
        out[:, : Ta * self.k] = dxp.view(B, Ta * self.k, self.d)
        return out.view(B * T2, self.d)

    def backward_hip(self, dy: torch.Tensor, stash: dict, accumulate: bool):
        s, p = self.store, self.prefix
        xp, h = stash.pop("proj")
        M = dy.shape[0]
        Mp = round_up(M, 64)
        dyT = ops.transpose(dy, Rp=Mp)
        hT = ops.transpose(h, Rp=Mp)
        ops.gemm_nt(dyT, hT, out=s.grad_view(p + "linear2.weight"), accumulate=accumulate)
        ops.colsum(dy, s.grad_view(p + "linear2.bias"), accumulate=accumulate)
        dh = ops.gemm_nt(dy, self.w2T)
        ops.relu_bwd_(dh, h)
        dhT = ops.transpose(dh, Rp=Mp)
        xT = ops.transpose(xp, Rp=Mp)
        ops.gemm_nt(dhT, xT, out=s.grad_view(p + "linear1.weight"), accumulate=accumulate)
        ops.colsum(dh, s.grad_view(p + "linear1.bias"), accumulate=accumulate)
        if self.need_dx:
            dxp = ops.gemm_nt(dh, self.w1T)
            return self._unstack(dxp, stash) if "proj_shape" in stash else dxp     # (forward_rows: the caller owns the row layout)
        stash.pop("proj_shape", None)
        return None

    def forward(self, x):
        return self.forward_hip(x, None)


class HipProjectorCov1d(nn.Module):
    """EncoderProjectorCov1d (src/slam_llm/models/projector.py:29-49): Conv1d(d, d, kernel = stride = k) over time, ReLU,
    Linear(d, 2048), ReLU, Linear(2048, llm_dim).  kernel == stride with no padding makes the conv a Linear over the
    k-frame stack, so it runs on the same GEMM as the concat projector: Wc[co, j*d + ci] = conv.weight[co, ci, j]
    (re-packed from the fp32 master at refresh; its gradient is un-packed into the reference's [co, ci, j] layout)."""

    def __init__(self, cfg: dict, store: TrainableStore, prefix="encoder_projector."):
        super().__init__()
        self.k, self.d, self.hid, self.dl = cfg["ds_rate"], cfg["enc_dim"], cfg["proj_hidden"], cfg["llm_dim"]
        self.store, self.prefix = store, prefix
        assert self.d % 64 == 0 and self.hid % 64 == 0 and self.dl % 64 == 0
        store.reserve(prefix + "conv1d.weight", (self.d, self.d, self.k))
        store.reserve(prefix + "conv1d.bias", (self.d,))
        store.reserve(prefix + "linear1.weight", (self.hid, self.d))
        store.reserve(prefix + "linear1.bias", (self.hid,))
        store.reserve(prefix + "linear2.weight", (self.dl, self.hid))
        store.reserve(prefix + "linear2.bias", (self.dl,))
        self.conv1d, self.linear1, self.linear2 = nn.Module(), nn.Module(), nn.Module()
        self.need_dx = False   # True when the encoder is trainable: backward_hip then returns dL/d(encoder output)

    _unstack = HipProjectorConcat._unstack

    def bind(self):
        s, p = self.store, self.prefix
        for m, n in ((self.conv1d, "conv1d"), (self.linear1, "linear1"), (self.linear2, "linear2")):
            m.weight, m.bias = s.params[p + n + ".weight"], s.params[p + n + ".bias"]

    def refresh(self):
        s, p = self.store, self.prefix
        d, k = self.d, self.k
        # frame-major packing of the conv taps: [co, ci, j] -> [co, j*d + ci] (one strided copy of 5*d^2 elements)
        # (all four updated in place after the first call: see ops.transpose_into)
        taps = s.bf16_view(p + "conv1d.weight").permute(0, 2, 1).reshape(d, k * d)
        if getattr(self, "wc", None) is None:
            self.wc = taps.contiguous()
        else:
            self.wc.copy_(taps)
        self.w1T = ops.transpose_into(getattr(self, "w1T", None), s.bf16_view(p + "linear1.weight"), self.hid)  # [d, hid]
        self.w2T = ops.transpose_into(getattr(self, "w2T", None), s.bf16_view(p + "linear2.weight"), self.dl)   # [hid, dl]
        self.wcT = ops.transpose_into(getattr(self, "wcT", None), self.wc, d) if self.need_dx else None          # [k*d, d]: dL/d(k-frame stack) = dc . Wc

    def forward_hip(self, enc: torch.Tensor, stash: Optional[dict]):
        """enc [B, T2, d] bf16 -> [B, T2 // k, dl] bf16"""
        B, T2, d = enc.shape
        Ta = T2 // self.k
        xp = enc[:, : Ta * self.k, :]
        if T2 % self.k:
            xp = xp.contiguous()
        if stash is not None:
            stash["proj_shape"] = (B, T2)
        return self.forward_rows(xp.reshape(B * Ta, self.k * d), stash).view(B, Ta, self.dl)

    def forward_rows(self, xp: torch.Tensor, stash: Optional[dict]):
        """xp [rows, k*d] bf16 k-frame stacked rows -> [rows, dl]"""
        s, p = self.store, self.prefix
        c = ops.gemm_nt(xp, self.wc, bias=s.master_view(p + "conv1d.bias"), act=ACT_RELU)
        h = ops.gemm_nt(c, s.bf16_view(p + "linear1.weight"), bias=s.master_view(p + "linear1.bias"), act=ACT_RELU)
        y = ops.gemm_nt(h, s.bf16_view(p + "linear2.weight"), bias=s.master_view(p + "linear2.bias"))
        if stash is not None:
            stash["proj"] = (xp, c, h)
        return y

    def backward_hip(self, dy: torch.Tensor, stash: dict, accumulate: bool):
        s, p = self.store, self.prefix
        xp, c, h = stash.pop("proj")
        M = dy.shape[0]
        Mp = round_up(M, 64)
        ops.gemm_nt(ops.transpose(dy, Rp=Mp), ops.transpose(h, Rp=Mp), out=s.grad_view(p + "linear2.weight"),
                    accumulate=accumulate)
        ops.colsum(dy, s.grad_view(p + "linear2.bias"), accumulate=accumulate)
        dh = ops.gemm_nt(dy, self.w2T)
        ops.relu_bwd_(dh, h)
        ops.gemm_nt(ops.transpose(dh, Rp=Mp), ops.transpose(c, Rp=Mp), out=s.grad_view(p + "linear1.weight"),
                    accumulate=accumulate)
        ops.colsum(dh, s.grad_view(p + "linear1.bias"), accumulate=accumulate)
        dc = ops.gemm_nt(dh, self.w1T)
        ops.relu_bwd_(dc, c)
        dwc = ops.gemm_nt(ops.transpose(dc, Rp=Mp), ops.transpose(xp, Rp=Mp), out_dtype=torch.float32)  # [d, k*d]
        g = s.grad_view(p + "conv1d.weight")                                  # reference layout [co, ci, j]
        dwc = dwc.view(self.d, self.k, self.d).permute(0, 2, 1)
        if accumulate:
            g.add_(dwc)
        else:
            g.copy_(dwc)
        ops.colsum(dc, s.grad_view(p + "conv1d.bias"), accumulate=accumulate)
        if self.need_dx:
            dxp = ops.gemm_nt(dc, self.wcT)
            return self._unstack(dxp, stash) if "proj_shape" in stash else dxp     # (forward_rows: the caller owns the row layout)
        stash.pop("proj_shape", None)
        return None

    def forward(self, x):
        return self.forward_hip(x, None)


# ======================================================================================== llama + LoRA
def _attach(root: nn.Module, dotted: str, param: nn.Parameter):
    """register `param` under a dotted path, creating plain container modules on the way."""
    parts = dotted.split(".")
    m = root
    for seg in parts[:-1]:
        if seg not in m._modules:
            m.add_module(seg, nn.Module())
        m = m._modules[seg]
    m.register_parameter(parts[-1], param)


class HipLlamaLora(nn.Module):
    """Frozen Llama decoder + peft-style LoRA adapters (HF LlamaForCausalLM under peft's LoraModel as built at
    src/slam_llm/models/slam_model.py:118-221); parameter names follow peft 0.6.0."""

    ATTN = ("q_proj", "k_proj", "v_proj", "o_proj")

    def __init__(self, cfg: dict, store: TrainableStore, device, prefix="llm."):
        super().__init__()
        self.cfg, self.store, self.device_, self.prefix = cfg, store, device, prefix
        d, Hq, Hkv, D, Fd = cfg["llm_dim"], cfg["llm_heads"], cfg["llm_kv_heads"], cfg["llm_head_dim"], cfg["llm_ffn"]
        self.lora_p = float(cfg.get("lora_dropout", 0.0) or 0.0)
        self._drop_calls = 0
        self.lm_head_chunk_rows = None   # None: rows per lm_head/CE chunk derived from a 1 GB bf16 logits buffer (tests override)
        # static upper bound on the rows that carry a label (None: the exact count is read back from the device each step): with it the
        # training forward has no host round trip and static shapes -- train.GraphedTrainStep sets it; `last_label_count` (device int32)
        # is the real count of the last forward, for the caller's overflow check
        self.label_rows_cap = None
        self.last_label_count = None
        self.layers = []
        targets = tuple(cfg.get("lora_targets") or ())
        r, alpha = cfg["lora_r"], cfg["lora_alpha"]
        P = prefix + "base_model.model."
        # reserve in backward-production order: last layer first
        for i in reversed(range(cfg["llm_layers"])):
            L = SimpleNamespace()
            L.qkv = FusedLinear(d, [("q_proj", Hq * D), ("k_proj", Hkv * D), ("v_proj", Hkv * D)], device)
            L.o = FusedLinear(Hq * D, [("o_proj", d)], device)
            L.gu = FusedLinear(d, [("gate_proj", Fd), ("up_proj", Fd)], device)
            L.down = FusedLinear(Fd, [("down_proj", d)], device)
            for fl in (L.down, L.gu, L.o, L.qkv):
                mine = [(n, rows) for n, rows in fl.parts if n in targets]
                for n, rows in mine:  # all A of a group first (contiguous A_cat), then the B's
                    mod = "self_attn." if n in self.ATTN else "mlp."
                    store.reserve(f"{P}model.layers.{i}.{mod}{n}.lora_A.default.weight", (r, fl.K))
                for n, rows in mine:
                    mod = "self_attn." if n in self.ATTN else "mlp."
                    base = f"{P}model.layers.{i}.{mod}{n}."
                    store.reserve(base + "lora_B.default.weight", (rows, r))
                    fl.add_lora(n, r, alpha, base + "lora_A.default.weight", base + "lora_B.default.weight")
            self.layers.insert(0, L)
        self._rope = {}

    # ---- weights -------------------------------------------------------------------------------------
    def bind(self):
        for name, p in self.store.params.items():
            if name.startswith(self.prefix):
                _attach(self, name[len(self.prefix):], p)

    def load(self, W: Dict[str, torch.Tensor]):
        cfg, dev = self.cfg, self.device_
        P = self.prefix + "base_model.model."
        bf = lambda t: t.to(device=dev, dtype=torch.bfloat16)  # noqa: E731
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        self.embed = bf(W[P + "model.embed_tokens.weight"]).contiguous()
        for i, L in enumerate(self.layers):
            p = f"{P}model.layers.{i}."
            for fl in (L.qkv, L.o, L.gu, L.down):
                for n, _ in fl.parts:
                    mod = "self_attn." if n in self.ATTN else "mlp."
                    fl.set_base(n, bf(W[p + mod + n + ".weight"]))
                fl.finalize()
            L.ln1, L.ln2 = f32(W[p + "input_layernorm.weight"]), f32(W[p + "post_attention_layernorm.weight"])
        self.norm_w = f32(W[P + "model.norm.weight"])
        self.lm_head = bf(W[P + "lm_head.weight"]).contiguous()
        self.lm_headT = ops.transpose(self.lm_head, Rp=self.lm_head.shape[0])
        return self

    def init_random(self, seed: int = 42):
        cfg, dev = self.cfg, self.device_
        d, Fd, V = cfg["llm_dim"], cfg["llm_ffn"], cfg["vocab"]
        g = torch.Generator(device=dev).manual_seed(seed)
        rn = lambda *s, std=0.02: (torch.randn(*s, generator=g, device=dev) * std)  # noqa: E731
        self.embed = rn(V, d, std=1.0).to(torch.bfloat16)
        for L in self.layers:
            for fl in (L.qkv, L.o, L.gu, L.down):
                for n, rows in fl.parts:
                    fl.set_base(n, rn(rows, fl.K, std=fl.K ** -0.5).to(torch.bfloat16))
                fl.finalize()
            L.ln1, L.ln2 = 1 + rn(d, std=0.1), 1 + rn(d, std=0.1)
        self.norm_w = 1 + rn(d, std=0.1)
        self.lm_head = rn(V, d, std=d ** -0.5).to(torch.bfloat16)
        self.lm_headT = ops.transpose(self.lm_head, Rp=V)
        return self

    def refresh(self):
        for L in self.layers:
            for fl in (L.qkv, L.o, L.gu, L.down):
                fl.refresh(self.store)

    def rope(self, T: int):
        if T not in self._rope:
            from .host_tables import rope_tables
            cos, sin = rope_tables(T, self.cfg["llm_head_dim"], self.cfg["rope_theta"])
            self._rope[T] = (cos.to(self.device_), sin.to(self.device_))
        return self._rope[T]

    # ---- forward -------------------------------------------------------------------------------------
    def forward_hip(self, h: torch.Tensor, B: int, T: int, key_mask: torch.Tensor, targets, n_valid,
                    train: bool, return_logits: bool, packed=None, label_count=None):
        """h [B*T, d] bf16 (consumed).  Returns (out2 = [loss, acc] device tensor or None, logits or None, stash).
        label_count = (event, pinned int32[1]): n_valid on its way to the host (see LM_HEAD_LABEL_ROWS)."""
        cfg, st = self.cfg, self.store
        d, Hq, Hkv, D, Fd, V = (cfg["llm_dim"], cfg["llm_heads"], cfg["llm_kv_heads"], cfg["llm_head_dim"],
                                cfg["llm_ffn"], cfg["vocab"])
        M = B * T
        eps = cfg["rms_eps"]
        # packed = (positions, seg_lo, seg_hi, max_len): B = 1, sequences concatenated along T without pad tokens
        positions, seg = (packed[0], (packed[1], packed[2])) if packed is not None else (None, None)
        cos, sin = self.rope(packed[3] if packed is not None else T)
        scale = D ** -0.5
        stash = {"layers": [], "B": B, "T": T, "key_mask": key_mask, "packed": packed} if train else None
        use_drop = self.training and self.lora_p > 0.0
        seed = torch.initial_seed() if use_drop else 0

        def drop_for(fl):
            """fresh (p, seed, offset) per adapted group and forward call; offsets never overlap (stride 2^40)"""
            if not (use_drop and fl.adapters):
                return None
            self._drop_calls += 1
            return (self.lora_p, seed, self._drop_calls << 40)

        n_lab, rows, tsel_rows, inv, prune = M, None, None, None, False
        self.last_label_count = None
        if targets is not None and train and not return_logits and (label_count is not None or self.label_rows_cap):
            # which rows carry a label: a device-side selection (slam_label_rows: row list, their targets, the inverse map, the count).
            # Its size is either the caller's STATIC bound (`label_rows_cap`: no host round trip at all -- what a captured step needs;
            # positions past the real count are zero rows with an ignored target, which change nothing) or the exact count read
            # back through pinned memory (an event wait that completed long ago: the count left at the start of the forward).
            cap = None
            if self.label_rows_cap:
                cap = min(int(self.label_rows_cap), M)
            else:
                label_count[0].synchronize()
                n_host = int(label_count[1][0])
                cap = n_host if 0 < n_host < M else None
            if cap is not None:
                n_lab = cap
                rows, tsel_rows, inv, self.last_label_count = ops.label_rows(targets, cap)
                LL = self.layers[-1]
                prune = LAST_LAYER_LABEL_ROWS and not (use_drop and bool(LL.o.adapters or LL.gu.adapters or LL.down.adapters))
        for L in self.layers:
            x1 = L.qkv.new_input(M, for_forward=True)
            _, rstd1 = ops.rmsnorm_fwd(h, L.ln1, eps, out=x1[:, :d])
            dq_, do_, dg_, dd_ = drop_for(L.qkv), drop_for(L.o), drop_for(L.gu), drop_for(L.down)
            qkv = L.qkv.forward(x1, st, drop=dq_)
            ops.rope_inplace(qkv, 0, B, T, Hq + Hkv, D, cos, sin, positions=positions)     # q and k heads: adjacent columns, one launch
            o_ext = L.o.new_input(M)
            _, lse = ops.attn_fwd(qkv[:, : Hq * D], qkv[:, Hq * D: (Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:], B, T, Hq, Hkv, D, True, scale,
                                  key_mask=key_mask, want_lse=train, out=o_ext[:, : Hq * D], seg=seg)
            pruned = prune and L is self.layers[-1]
            Mr, o_in, h_res = M, o_ext, h
            if pruned:      # behind the last attention only the labelled rows matter
                Mr = self._last_pruned_rows = n_lab
                o_in = L.o.new_input(Mr)
                ops.gather_rows(o_ext[:, : Hq * D], rows, out=o_in[:, : Hq * D])
                h_res = ops.gather_rows(h, rows)
            h_mid = L.o.forward(o_in, st, residual=h_res, drop=do_)
            x2 = L.gu.new_input(Mr)
            _, rstd2 = ops.rmsnorm_fwd(h_mid, L.ln2, eps, out=x2[:, :d])
            hh = L.down.new_input(Mr)
            gu_il = train and L.gu.Wil is not None and ops.gemm_swiglu_supported(Mr, 2 * Fd, d, x2.stride(0), L.gu.Wil.stride(0))
            if gu_il:   # one launch: [gate64 | up64]-block stash for the backward + h = silu(gate) * up
                gu = torch.empty((Mr, 2 * Fd), dtype=torch.bfloat16, device=h.device)
                ops.gemm_swiglu(x2[:, :d], L.gu.Wil, gu, hh[:, :Fd])
            else:
                gu = L.gu.forward(x2, st, drop=dg_)
                ops.swiglu_fwd(gu, out=hh[:, :Fd])
            h_out = L.down.forward(hh, st, residual=h_mid, drop=dd_)
            if train:
                stash["layers"].append(dict(h=h, rstd1=rstd1, x1=x1 if L.qkv.adapters else None, qkv=qkv,
                                            o=o_in, o_full=o_ext if pruned else None, inv=inv if pruned else None,
                                            lse=lse, h_mid=h_mid, rstd2=rstd2,
                                            x2=x2 if L.gu.adapters else None, gu=gu, gu_il=gu_il,
                                            hh=hh if L.down.adapters else None, drops=(dq_, do_, dg_, dd_)))
            h = h_out
        hN, rstdN = ops.rmsnorm_fwd(h, self.norm_w, eps)
        logits_full = torch.empty((M, V), dtype=torch.bfloat16, device=h.device) if return_logits else None
        out2 = None
        if targets is not None:
            hsel, tsel = hN, targets
            if rows is not None:
                hsel, tsel = (hN if prune else ops.gather_rows(hN, rows)), tsel_rows
            row_loss = torch.empty((n_lab,), dtype=torch.float32, device=h.device)
            row_ok = torch.empty((n_lab,), dtype=torch.int32, device=h.device)
            dhN = torch.empty((n_lab, d), dtype=torch.bfloat16, device=h.device) if train else None
            Rc = self.lm_head_chunk_rows or max(256, min(n_lab, ((1 << 29) // V) // 256 * 256))
            chunk = torch.empty((min(Rc, n_lab), V), dtype=torch.bfloat16, device=h.device)
            for r0 in range(0, n_lab, Rc):
                r1 = min(n_lab, r0 + Rc)
                lg = chunk[: r1 - r0]
                ops.gemm_nt(hsel[r0:r1], self.lm_head, out=lg)
                if return_logits:
                    logits_full[r0:r1].copy_(lg)
                ops.ce_fwd_bwd(lg, tsel[r0:r1], n_valid, row_loss[r0:r1], row_ok[r0:r1], write_grad=train)
                if train:
                    ops.gemm_nt(lg, self.lm_headT, out=dhN[r0:r1])
            out2 = ops.ce_finalize(row_loss, row_ok, n_valid)
            if train:
                if rows is not None and not prune:    # back to the [M, d] layout: the rows without a label carry a zero gradient
                    dhN = ops.gather_rows(dhN, inv)
                stash["final"] = dict(h=h, rstdN=rstdN, dhN=dhN)
        elif return_logits:
            Rc = 4096
            for r0 in range(0, M, Rc):
                r1 = min(M, r0 + Rc)
                ops.gemm_nt(hN[r0:r1], self.lm_head, out=logits_full[r0:r1])
        return out2, logits_full, stash

    # ---- decode (generate) ---------------------------------------------------------------------------
    # KV layout (all bf16, row-major [.., Hkv*D]):
    #   prompt     Kp/Vp [layers, B, T, Hkv*D]   written once by prefill, SHARED by the beams of a batch item
    #   generated  Kg/Vg [layers, R, G, Hkv*D]   R = B*beams rows, G = max_new_tokens slots, append-only
    #   anc        [R, G] int32                  physical row holding slot j of hypothesis r's history
    # Beam search re-ranks hypotheses every step; instead of re-ordering the cache (HF: index_select over every
    # layer's K and V), only `anc` is gathered -- a [R, G] int table shared by all layers.
    def _mlp_block(self, L, h_mid):
        cfg, st = self.cfg, self.store
        d, Fd = cfg["llm_dim"], cfg["llm_ffn"]
        M = h_mid.shape[0]
        x2 = L.gu.new_input(M)
        ops.rmsnorm_fwd(h_mid, L.ln2, cfg["rms_eps"], out=x2[:, :d])
        gu = L.gu.forward(x2, st)
        hh = L.down.new_input(M)
        ops.swiglu_fwd(gu, out=hh[:, :Fd])
        return L.down.forward(hh, st, residual=h_mid)

    def _next_logits(self, h_last: torch.Tensor) -> torch.Tensor:
        hN, _ = ops.rmsnorm_fwd(h_last, self.norm_w, self.cfg["rms_eps"])
        return ops.gemm_nt(hN, self.lm_head, out_dtype=torch.float32)

    @torch.no_grad()
    def prefill(self, h: torch.Tensor, B: int, T: int, attention_mask: torch.Tensor, max_new_tokens: int, beams: int = 1):
        """Run the prompt (h [B*T, d] bf16, LEFT padded) and build the KV cache for B*beams hypotheses.  Rotary
        positions follow HF generate(): cumsum(attention_mask) - 1 per row, NOT arange (which the training forward
        uses).  Returns (next-token logits [B, V] fp32, cache)."""
        cfg, st, dev = self.cfg, self.store, h.device
        d, Hq, Hkv, D = cfg["llm_dim"], cfg["llm_heads"], cfg["llm_kv_heads"], cfg["llm_head_dim"]
        M, R, G, Ln, eps = B * T, B * beams, max_new_tokens, len(self.layers), cfg["rms_eps"]
        am = attention_mask.to(device=dev, dtype=torch.int32)
        if not bool((am[:, 1:] >= am[:, :-1]).all()) or not bool(am[:, -1].all()):
            raise ValueError("generate(): the prompt batch must be left padded (speech_dataset.py:216-273 inference collator)")
        positions = (am.cumsum(-1) - 1).clamp_(min=0).to(torch.int32).contiguous()
        n_real = am.sum(-1).to(torch.int32)
        cos, sin = self.rope(T + G)
        Tp64 = round_up(T, 64)
        key_mask = torch.zeros((B, Tp64), dtype=torch.uint8, device=dev)
        key_mask[:, :T] = am.to(torch.uint8)
        kv = dict(dtype=torch.bfloat16, device=dev)
        cache = SimpleNamespace(
            B=B, R=R, beams=beams, T=T, G=G, n=0,
            Kp=torch.empty((Ln, B, T, Hkv * D), **kv), Vp=torch.empty((Ln, B, T, Hkv * D), **kv),
            Kg=torch.zeros((Ln, R, G, Hkv * D), **kv), Vg=torch.zeros((Ln, R, G, Hkv * D), **kv),
            anc=torch.arange(R, dtype=torch.int32, device=dev)[:, None].repeat(1, G).contiguous(),
            start=(T - n_real).to(torch.int32).contiguous(),
            next_pos=n_real.repeat_interleave(beams).contiguous(),
            n_dev=torch.zeros(1, dtype=torch.int32, device=dev), tok=torch.zeros(R, dtype=torch.int64, device=dev),
            graph=None, logits=None)
        for li, L in enumerate(self.layers):
            x1 = L.qkv.new_input(M)
            ops.rmsnorm_fwd(h, L.ln1, eps, out=x1[:, :d])
            qkv = L.qkv.forward(x1, st)
            ops.rope_inplace(qkv, 0, B, T, Hq + Hkv, D, cos, sin, positions=positions)
            cache.Kp[li].copy_(qkv[:, Hq * D:(Hq + Hkv) * D].view(B, T, Hkv * D))
            cache.Vp[li].copy_(qkv[:, (Hq + Hkv) * D:].view(B, T, Hkv * D))
            o_ext = L.o.new_input(M)
            ops.attn_fwd(qkv[:, : Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:], B, T, Hq, Hkv, D, True, D ** -0.5,
                         key_mask=key_mask, want_lse=False, out=o_ext[:, : Hq * D])
            h = self._mlp_block(L, L.o.forward(o_ext, st, residual=h))
        last = h.view(B, T, -1)[:, T - 1].contiguous()
        return self._next_logits(last), cache

    @torch.no_grad()
    def reorder_cache(self, cache, rows: torch.Tensor):
        """hypothesis r continues old hypothesis rows[r] (same batch item): gather the ancestor table only."""
        cache.anc.copy_(cache.anc.index_select(0, rows))
        return cache

    def _decode_body(self, cache) -> torch.Tensor:
        """one decode step over static buffers (cache.tok, cache.next_pos, cache.n_dev, cache.anc): every launch
        reads the step index from device memory, so the sequence can be captured once in a HIP graph and replayed.
        7 launches per layer for <= 64 hypotheses: rmsnorm, qkv (+ LoRA A rows), attention (LoRA delta + RoPE + KV
        append fused), o_proj (+ residual), rmsnorm, gate_up (+ SwiGLU), down_proj (+ residual)."""
        cfg, st = self.cfg, self.store
        d, Hq, Hkv, D, Fd = cfg["llm_dim"], cfg["llm_heads"], cfg["llm_kv_heads"], cfg["llm_head_dim"], cfg["llm_ffn"]
        R, eps = cache.R, cfg["rms_eps"]
        skinny = R <= ops.SKINNY_MAX_M
        h = self.embed.index_select(0, cache.tok)
        cos, sin = self.rope(cache.T + cache.G)
        for li, L in enumerate(self.layers):
            fq = L.qkv
            if skinny and fq.adapters:
                # u = x A^T comes out of the same launch as q|k|v (A stacked under W); the attention kernel adds
                # u . (scale B)^T to the new token's q/k/v (peft: base(x) + scale * B(A(x)))
                x1, _ = ops.rmsnorm_fwd(h, L.ln1, eps)
                qkv = torch.empty((R, fq.N + fq.sum_r), dtype=torch.bfloat16, device=h.device)
                ops.gemm_skinny(x1, fq.Wext[:, : fq.K], qkv, b2=fq.a_cat(st))
                lora_b, lora_r = fq.Wext[:, fq.K:], fq.sum_r
            else:
                x1 = fq.new_input(R)
                ops.rmsnorm_fwd(h, L.ln1, eps, out=x1[:, :d])
                qkv, lora_b, lora_r = fq.forward(x1, st), None, 0
            o_ext = L.o.new_input(R)
            ops.attn_decode(qkv, lora_b, lora_r, cos, sin, cache.next_pos, cache.Kp[li], cache.Vp[li], cache.start,
                            cache.Kg[li], cache.Vg[li], cache.anc, cache.n_dev, cache.n, cache.beams, Hq, Hkv, D,
                            D ** -0.5, o_ext)
            h_mid = L.o.forward(o_ext, st, residual=h)
            x2 = L.gu.new_input(R)
            ops.rmsnorm_fwd(h_mid, L.ln2, eps, out=x2[:, :d])
            hh = L.down.new_input(R)
            if skinny:
                L.gu.forward_swiglu(x2, st, hh[:, :Fd])
            else:
                ops.swiglu_fwd(L.gu.forward(x2, st), out=hh[:, :Fd])
            h = L.down.forward(hh, st, residual=h_mid)
        logits = self._next_logits(h)
        cache.n_dev += 1
        cache.next_pos += 1
        return logits

    @torch.no_grad()
    def decode_step(self, tokens: torch.Tensor, cache):
        """append one token per hypothesis; returns next-token logits [R, V] fp32 (a buffer re-used by the next step).
        ~330 short launches per step: captured into a HIP graph on the first step and replayed afterwards
        (cfg['decode_graph'], default on; per-kernel timing via ops.TIMER runs eagerly)."""
        if cache.n >= cache.G:
            raise RuntimeError("KV cache capacity exhausted (max_new_tokens)")
        cache.tok.copy_(tokens)
        if not self.cfg.get("decode_graph", True) or ops.TIMER is not None:
            logits = self._decode_body(cache)
        else:
            if cache.graph is None:
                self.rope(cache.T + cache.G)   # tables must exist before capture (host -> device copy)
                cache.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(cache.graph):
                    cache.logits = self._decode_body(cache)
            cache.graph.replay()
            logits = cache.logits
        cache.n += 1
        return logits

    # ---- backward ------------------------------------------------------------------------------------
    def backward_hip(self, stash: dict, grad_scale: Optional[torch.Tensor], accumulate: bool, on_layer_done=None):
        cfg, st = self.cfg, self.store
        d, Hq, Hkv, D, Fd = cfg["llm_dim"], cfg["llm_heads"], cfg["llm_kv_heads"], cfg["llm_head_dim"], cfg["llm_ffn"]
        B, T, key_mask = stash["B"], stash["T"], stash["key_mask"]
        packed = stash.get("packed")
        cos, sin = self.rope(packed[3] if packed is not None else T)
        rope = (cos, sin, packed[0]) if packed is not None else (cos, sin)
        seg = (packed[1], packed[2]) if packed is not None else None
        scale = D ** -0.5
        f = stash.pop("final")
        dh = ops.rmsnorm_bwd(f["h"], f["rstdN"], self.norm_w, f["dhN"], grad_scale=grad_scale)
        del f
        # Adapter-gradient products (dA, dB grams + their fixed-order reduces: ~3 ms of HBM-bound launches per C3 step) are off the dX
        # critical path: with LORA_SIDE_STREAM they run on a second stream, under the NEXT layer's GEMMs (whose partial last rounds leave
        # CUs idle), and are joined one layer late -- the main stream waits for layer li's products at the end of layer li - 1, and only
        # then are layer li's gradient-sync hooks fired.  Same kernels, same operands, same fixed-order reductions: bit-identical.
        side = _side_stream(dh.device) if LORA_SIDE_STREAM else None
        pending = []        # [(done event, tensors kept alive until the main stream has waited for it)]
        announced = [len(self.layers)]      # layers >= announced[0] have had their on_layer_done call

        def defer(fn, keep):
            main = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(main)
            side.wait_event(ready)
            with torch.cuda.stream(side):
                fn()
            done = torch.cuda.Event()
            done.record(side)
            pending.append((done, keep))

        def join_and_announce(down_to: int):
            """the main stream waits for all side work queued so far, then layers announced[0] - 1 ... down_to get their hook call"""
            main = torch.cuda.current_stream()
            while pending:
                done, _keep = pending.pop(0)
                main.wait_event(done)
            while announced[0] > down_to:
                announced[0] -= 1
                if on_layer_done is not None:
                    on_layer_done(announced[0])

        dfr = defer if side is not None else None
        for li in reversed(range(len(self.layers))):
            L, S = self.layers[li], stash["layers"][li]
            dq_, do_, dg_, dd_ = S["drops"]
            if L.down.adapters or not FUSE_SWIGLU_BWD or S["gu_il"]:
                d_hh = L.down.backward(dh, S["hh"], st, accumulate, drop=dd_, defer=dfr)
                dgu = ops.swiglu_bwd(S["gu"], d_hh[:, :Fd], interleaved=S["gu_il"])
                del d_hh
            else:
                # frozen down_proj: dL/dh = dy . W_down never leaves the GEMM -- its epilogue applies the SwiGLU backward
                # against the stashed [gate | up] and writes dL/dgate | dL/dup directly
                dgu = torch.empty_like(S["gu"])
                ops.gemm_nt(dh, L.down.WextT, out=dgu[:, :Fd], act=ops.ACT_SWIGLU_BWD, residual=S["gu"])
            dx2 = L.gu.backward(dgu, S["x2"], st, accumulate, drop=dg_, defer=dfr)
            del dgu
            dh_mid = ops.rmsnorm_bwd(S["h_mid"], S["rstd2"], L.ln2, dx2[:, :d], dres=dh)
            del dx2
            if side is not None and li + 1 < len(self.layers):
                # the products of the layer ABOVE had this layer's MLP backward (the two largest GEMMs of a layer) to hide under
                join_and_announce(li + 1)
            do_ext = L.o.backward(dh_mid, S["o"], st, accumulate, drop=do_, defer=dfr)
            dO, o_attn = do_ext[:, : Hq * D], S["o"]
            if S["inv"] is not None:    # last layer ran behind its attention over the labelled rows only: back to every row (zeros elsewhere)
                dO, dh_mid, o_attn = ops.gather_rows(dO, S["inv"]), ops.gather_rows(dh_mid, S["inv"]), S["o_full"]
            qkv = S["qkv"]
            q2, k2, v2 = qkv[:, : Hq * D], qkv[:, Hq * D: (Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
            dqkv = torch.empty_like(qkv)
            ops.attn_bwd(q2, k2, v2, o_attn[:, : Hq * D], dO, S["lse"],
                         dqkv[:, : Hq * D], dqkv[:, Hq * D: (Hq + Hkv) * D], dqkv[:, (Hq + Hkv) * D:],
                         B, T, Hq, Hkv, D, True, scale, key_mask=key_mask, rope=rope, seg=seg)  # RoPE backward fused
            del do_ext, dO
            dx1 = L.qkv.backward(dqkv, S["x1"], st, accumulate, drop=dq_, defer=dfr)
            del dqkv
            dh = ops.rmsnorm_bwd(S["h"], S["rstd1"], L.ln1, dx1[:, :d], dres=dh_mid)
            del dx1, dh_mid
            stash["layers"][li] = None
            if side is None and on_layer_done is not None:
                on_layer_done(li)
        if side is not None:
            join_and_announce(0)
        return dh


# ======================================================================================== composite
class _SlamStep(torch.autograd.Function):
    """autograd entry: forward returned the loss computed by the HIP path; backward runs the HIP backward.

    Two ways of handing the gradients over:
      * flat-buffer mode (default, fast path): gradients are deposited in place into the flat grad buffer that `.grad`
        of every trainable parameter views; autograd sees no parameter gradients (returns None for them), gradient
        accumulation is an in-place add by the kernels, `GradSync` all-reduces buffer prefixes during the backward;
      * `model.autograd_params` (DDP-compatible): the trainable parameters are real inputs of this node and the backward
        returns views of a FRESH flat buffer as their gradients, so AccumulateGrad (and with it the reducer hooks of
        `torch.nn.parallel.DistributedDataParallel`, src/slam_llm/pipeline/finetune.py:181-184) fires for each of
        them; autograd adopts the views without a copy when `.grad` is None and adds into `.grad` otherwise."""

    @staticmethod
    def forward(ctx, anchor, model, stash, loss_value, *params):
        ctx.model, ctx.stash = model, stash
        ctx.n_params = len(params)
        return loss_value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        model, stash = ctx.model, ctx.stash
        ctx.stash = None
        grads = model._run_backward(stash, grad_out, as_autograd=ctx.n_params > 0)
        return (None, None, None, None) + tuple(grads)


class SlamHipModel(nn.Module):
    def __init__(self, cfg: dict, device, tokenizer=None, train_config=None, model_config=None, **kwargs):
        super().__init__()
        self.cfg = cfg
        self.device_ = torch.device(device)
        self.tokenizer = tokenizer
        self.train_config, self.model_config = train_config, model_config
        self.metric = kwargs.get("metric", "acc")
        self.store = TrainableStore(self.device_)
        self.encoder_name = cfg.get("encoder_name", "whisper")
        self.projector_name = cfg.get("projector", "linear")
        # train_config.freeze_encoder=false (models/slam_model.py:110-113): the encoder's parameters join the trainable store
        self.train_encoder = not bool(cfg.get("freeze_encoder", True))
        if self.train_encoder and self.encoder_name not in ("whisper", "hubert", "wavlm"):
            raise NotImplementedError("freeze_encoder=false is implemented for the Whisper, HuBERT and WavLM encoders (hand-written encoder "
                                      "backward) with the linear / cov1d-linear / q-former projectors")
        if self.encoder_name in ("hubert", "wavlm"):
            cfg["enc_dim"] = cfg["hub_dim"]
            if not self.train_encoder:
                self.encoder = (HipHubertEncoder if self.encoder_name == "hubert" else HipWavLMEncoder)(cfg, self.device_)
        elif not self.train_encoder:
            self.encoder = HipWhisperEncoder(cfg, self.device_)
        self.llm = HipLlamaLora(cfg, self.store, self.device_)          # reserves LoRA (last layer first)
        if self.projector_name == "q-former":
            from .qformer import HipProjectorQFormer
            self.encoder_projector = HipProjectorQFormer(cfg, self.store)
        elif self.projector_name == "cov1d-linear":
            self.encoder_projector = HipProjectorCov1d(cfg, self.store)
        elif self.projector_name != "linear":
            raise ValueError(f"unknown encoder_projector {self.projector_name!r} (linear | cov1d-linear | q-former)")
        else:
            self.encoder_projector = HipProjectorConcat(cfg, self.store)  # projector last = produced last in backward
        if self.train_encoder:   # ... except a trainable encoder, after it
            enc_cls = {"hubert": HipHubertEncoder, "wavlm": HipWavLMEncoder, "whisper": HipWhisperEncoder}[self.encoder_name]
            self.encoder = enc_cls(cfg, self.device_, store=self.store)
            self.encoder_projector.need_dx = True
        self.store.allocate()
        self.llm.bind()
        self.encoder_projector.bind()
        if self.train_encoder:
            self.encoder.bind()
        self._anchor = torch.zeros(1, device=self.device_, requires_grad=True)
        self._stale = True
        self.return_logits = None  # None: logits only in eval mode
        self.grad_hooks = []       # GradSync-like objects: .on_backward_begin(), .on_prefix(end_offset)
        self._always_refresh = True
        # True: trainable parameters are autograd inputs of the step and receive their gradients through AccumulateGrad
        # (what DistributedDataParallel needs); set by model_factory when train_config.enable_ddp is on
        self.autograd_params = bool(kwargs.get("autograd_params", False))

    # ---- weights -------------------------------------------------------------------------------------
    def load_weights(self, W: Dict[str, torch.Tensor], seed: int = 42):
        """W: {name: tensor} with the reference's state_dict names (fp32 or bf16, any device).  Trainable tensors that
        W does not carry (a fresh fine-tune from pretrained encoder/LLM weights) are initialised the way the reference's
        modules initialise themselves -- see `init_missing_trainables`."""
        self.encoder.load(W)
        self.llm.load(W)
        missing = []
        with torch.no_grad():
            for name, p in self.store.params.items():
                if self.train_encoder and name.startswith("encoder."):
                    continue   # a trainable encoder loaded its own parameters above (incl. the folded weight-norm tensor)
                if name in W:
                    p.copy_(W[name].to(self.device_, torch.float32))
                else:
                    missing.append(name)
        if missing:
            self.init_missing_trainables(missing, seed)
        self._stale = True
        return self

    @torch.no_grad()
    def init_missing_trainables(self, names, seed: int = 42):
        """Fresh-module initialisation of the reference's trainable parts (the flat store itself is zero-filled, which
        would leave the projector at a fixed point -- relu(0) kills every gradient but linear2.bias -- and LoRA with
        A = B = 0 untrainable):
          * projector nn.Linear / nn.Conv1d (src/slam_llm/models/projector.py:11-13,35-38): torch's reset_parameters,
            kaiming_uniform_(a=sqrt(5)) = U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for the weight and the same bound for the bias;
          * peft 0.6.0 LoRA: lora_A kaiming_uniform_(a=sqrt(5)), lora_B zeros;
          * Q-Former (projector.py:60-67): query ~ N(0, 1); Blip2QFormerModel._init_weights: Linear weights N(0, 0.02),
            biases 0, LayerNorm weight 1 / bias 0; the output nn.Linear by reset_parameters, nn.LayerNorm 1 / 0.
        Seeded from train_config.seed through a CPU generator (the same values on every rank)."""
        g = torch.Generator().manual_seed(int(seed))
        st = self.store
        shapes = {n: shape for n, shape, _ in st.entries}

        def uniform(shape, bound):
            return (torch.rand(shape, generator=g) * 2 - 1) * bound

        def fan_in_of(weight_name):
            sh = shapes[weight_name]
            return int(math.prod(sh[1:]))

        qformer_inner = "encoder_projector.qformer."
        for name in names:
            p, sh = st.params[name], shapes[name]
            if "lora_B" in name:
                val = torch.zeros(sh)
            elif "lora_A" in name:
                val = uniform(sh, 1.0 / math.sqrt(sh[1]))
            elif name == "encoder_projector.query":
                val = torch.randn(sh, generator=g)
            elif name.startswith(qformer_inner):
                if "LayerNorm" in name or "layernorm" in name:
                    val = torch.ones(sh) if name.endswith("weight") else torch.zeros(sh)
                elif name.endswith("bias"):
                    val = torch.zeros(sh)
                else:
                    val = torch.randn(sh, generator=g) * 0.02
            elif name == "encoder_projector.norm.weight":
                val = torch.ones(sh)
            elif name == "encoder_projector.norm.bias":
                val = torch.zeros(sh)
            elif name.endswith(".weight"):
                val = uniform(sh, 1.0 / math.sqrt(fan_in_of(name)))
            elif name.endswith(".bias"):
                val = uniform(sh, 1.0 / math.sqrt(fan_in_of(name[: -len("bias")] + "weight")))
            else:
                raise RuntimeError(f"no initialisation rule for trainable tensor {name}")
            p.copy_(val.to(self.device_, torch.float32))
        self._stale = True

    def init_random(self, seed: int = 42, lora_b_std: float = 0.02):
        self.encoder.init_random(seed)
        self.llm.init_random(seed + 1)
        g = torch.Generator(device=self.device_).manual_seed(seed + 2)
        with torch.no_grad():
            for name, p in self.store.params.items():
                if name.startswith("encoder."):
                    continue   # a trainable encoder initialised itself above
                if ("LayerNorm" in name or "layernorm" in name or name.endswith("norm.weight")) and name.endswith("weight"):
                    p.fill_(1.0)
                elif name.endswith("bias"):
                    p.normal_(0, 0.02, generator=g)
                elif "lora_B" in name:
                    p.normal_(0, lora_b_std, generator=g)
                else:
                    p.normal_(0, p.shape[-1] ** -0.5, generator=g)
        self._stale = True
        return self

    def mark_params_updated(self):
        """call after an optimizer step that bypassed SlamAdamW (e.g. torch.optim.AdamW on .parameters())."""
        self._stale = True

    def _refresh(self):
        self.store.refresh_bf16()
        self.refresh_derived()

    def refresh_derived(self):
        """everything computed FROM the bf16 copies of the trainable parameters (fused / transposed / re-laid-out operands);
        the fused optimizers call this after writing the bf16 copies themselves"""
        self.llm.refresh()
        self.encoder_projector.refresh()
        if self.train_encoder:
            self.encoder.refresh()
        self._stale = False

    def train(self, mode: bool = True):
        super().train(mode)
        return self

    def _apply(self, fn, *args, **kwargs):
        """`.cuda(local_rank)` of the reference's pipeline (finetune.py:181) is a no-op move; anything that would change
        the dtype or the device of the trainable parameters (`model.to(torch.bfloat16)` of the pure_bf16 route,
        finetune.py:154-155; `.cpu()`) would silently detach them from the flat master buffer the kernels read."""
        probe = fn(torch.empty(0, dtype=torch.float32, device=self.device_))
        if probe.device == self.device_ and probe.dtype == torch.bfloat16:
            # the pure_bf16 route (finetune.py:154-155 `model.to(torch.bfloat16)`): bf16 masters, see TrainableStore.to_pure_bf16
            self.store.to_pure_bf16()
            self._round_frozen_fp32()
            self._stale = True
            return self
        want = torch.bfloat16 if self.store.pure_bf16 else torch.float32
        probe = fn(torch.empty(0, dtype=want, device=self.device_))
        if probe.dtype != want or probe.device != self.device_:
            raise RuntimeError(f"SlamHipModel lives on {self.device_} with {'bf16' if self.store.pure_bf16 else 'fp32'} trainable masters (`.to(torch.bfloat16)` "
                               f"switches to bf16 masters once); cannot convert to {probe.dtype} on {probe.device}")
        if self.store.pure_bf16:
            return self     # (a no-op move: the parameters are views of the flat bf16 buffer and must stay that)
        return super()._apply(fn, *args, **kwargs)

    @torch.no_grad()
    def _round_frozen_fp32(self):
        """the other half of `model.to(torch.bfloat16)`: in the reference EVERY parameter becomes bf16, the frozen ones included -- the
        LayerNorm / RMSNorm weights and the biases this path keeps in fp32 (the kernels read them in fp32) take their bf16-representable
        values (the frozen matrices are bf16 already)."""
        flat = self.store.flat
        lo, hi = flat.data_ptr(), flat.data_ptr() + 4 * flat.numel()

        def rnd(t):
            if isinstance(t, torch.Tensor) and t.dtype == torch.float32 and not (lo <= t.data_ptr() < hi):
                t.copy_(t.to(torch.bfloat16))
        for t in getattr(self.encoder, "w", {}).values():
            rnd(t)
        if hasattr(self.encoder, "refold_query_bias"):
            self.encoder.refold_query_bias()
        for L in self.llm.layers:
            rnd(L.ln1); rnd(L.ln2)
        rnd(self.llm.norm_w)

    # ---- forward -------------------------------------------------------------------------------------
    def forward(self, input_ids=None, attention_mask=None, labels=None, **kwargs):
        """same contract as slam_model.forward (src/slam_llm/models/slam_model.py:283-407)."""
        dev = self.device_
        audio_mel = kwargs.get("audio_mel", None)
        audio = kwargs.get("audio", None)
        modality_mask = kwargs.get("modality_mask", None)
        if input_ids is None or input_ids.device.type != "cuda":
            raise RuntimeError("SlamHipModel.forward needs the batch resident in HBM (move it with .to(device) as the "
                               "reference's train loop does, utils/train_utils.py:101-111)")
        if self._stale or self._always_refresh:
            # params may have been updated by a foreign optimizer (torch.optim.AdamW on .parameters()) -- also right before an
            # eval-mode forward (the reference's evaluation() follows optimizer.step() directly, utils/train_utils.py:173-177);
            # refreshing is one cheap pass over the ~30 M trainable parameters.  SlamAdamW refreshes itself and turns this off.
            self._refresh()
        B, T = input_ids.shape
        train = torch.is_grad_enabled() and labels is not None
        stash = {} if train else None
        early_targets = label_count = None
        if train and LM_HEAD_LABEL_ROWS and self.llm.label_rows_cap:
            early_targets = ops.ce_targets(labels.contiguous())     # static bound: nothing goes to the host
        elif train and LM_HEAD_LABEL_ROWS:
            # the shifted targets and their count now, the count on its way to pinned host memory: the LLM head reads it ~a forward later
            early_targets = ops.ce_targets(labels.contiguous())
            if getattr(self, "_label_count_host", None) is None:
                self._label_count_host = torch.empty(1, dtype=torch.int32).pin_memory()
            self._label_count_host.copy_(early_targets[1], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            label_count = (ev, self._label_count_host)

        hub_pad = None
        if self.encoder_name in ("hubert", "wavlm"):
            # raw-waveform encoders (slam_model.py:335-341 HuBERT, :333-334 WavLM: both get `1 - audio_mask` as the padding mask)
            if audio is None:
                raise RuntimeError(f"{self.encoder_name} encoder needs the raw `audio` batch key")
            # valid samples per clip as host ints: the collator's python list when present (no sync), else audio_len / audio_mask
            nv = kwargs.get("audio_len_list", None)
            if nv is None and kwargs.get("audio_len", None) is not None:
                nv = kwargs["audio_len"].tolist()
            if nv is None and kwargs.get("audio_mask", None) is not None:
                nv = (kwargs["audio_mask"] > 0).sum(1).tolist()
            hub_pad = None
            hub_train = self.train_encoder and train
            if nv is not None and min(nv) < audio.shape[1]:
                # ragged batch: the reference hands fairseq `padding_mask = 1 - audio_mask` (slam_model.py:336)
                nvi = [int(n) for n in nv]
                enc = self.encoder.forward_train(audio.float(), stash, nvi) if hub_train else self.encoder.forward_wav(audio.float(), nvi)
                keep = self.encoder.valid_frames(audio.shape[1], nv)
                # fairseq's frame padding mask: 1 = PADDING (frames kf .. of clip b)
                hub_pad = (torch.arange(enc.shape[1])[None, :] >= torch.tensor(keep)[:, None]).to(torch.float32).to(dev, non_blocking=True)
            else:
                enc = self.encoder.forward_train(audio.float(), stash) if hub_train else self.encoder.forward_wav(audio.float())
        else:
            if audio_mel is None:
                if audio is None:
                    raise RuntimeError("batch carries neither audio_mel nor audio")
                # GPU log-mel front end (replaces the CPU DataLoader mel of speech_dataset.py:101-103)
                if self.cfg.get("pad_or_trim", True):   # reference default (aispeech_asr_config.py:106; unconditional in speech_dataset.py:101)
                    with trace.phase("mel"):
                        audio_mel = ops.logmel(audio.float(), self.cfg["n_mels"], n_valid=kwargs.get("audio_len", None))
                else:                                   # ragged clips: mel over each clip's own length, zero padded to the batch max
                    alen = kwargs.get("audio_len", None)
                    nmax = min(480000, round_up(int(audio.shape[1]), 160))
                    audio_mel = ops.logmel(audio.float(), self.cfg["n_mels"], n_samples=nmax, n_valid=alen, per_clip=True)
            n_frames = self._ragged_frames(audio_mel, kwargs) if self.cfg.get("varlen_encoder", False) else None
            if n_frames is not None:
                proj = self._encode_project_ragged(audio_mel.float().contiguous(), n_frames, stash)
                enc = None
            else:
                mel_c = audio_mel.float().contiguous()
                with trace.phase("encoder"):
                    enc = self.encoder.forward_train(mel_c, stash) if (self.train_encoder and train) else self.encoder.forward_btc(mel_c)
        if enc is None:
            pass
        elif self.projector_name == "q-former":
            # audio_mel_post_mask is consumed only by this branch (slam_model.py:354-355); None = attend to all
            if self.encoder_name in ("whisper", "wavlm"):   # (the WavLM branch never produces a post mask, slam_model.py:333-334)
                pmask = kwargs.get("audio_mel_post_mask", None)
            elif hub_pad is None:
                pmask = None                    # nothing padded: fairseq returns no mask, the Q-Former attends to everything
            elif self.cfg.get("hubert_qformer_mask_fix", False):
                pmask = 1.0 - hub_pad           # corrected: attend to the REAL frames (++model_config.hubert_qformer_mask_fix=true)
            else:
                # reference behaviour: fairseq's padding mask (1 = padding) is handed to the Q-Former as
                # encoder_attention_mask (1 = attend) WITHOUT inversion (slam_model.py:338-341 vs the av_hubert branch :346): a
                # padded clip cross-attends to its padding frames only; a clip without padding has an all-zero mask, which HF
                # turns into one constant additive bias = ordinary attention over all frames.
                pmask = torch.where(hub_pad.sum(1, keepdim=True) > 0, hub_pad, torch.ones_like(hub_pad))
            with trace.phase("projector"):
                proj = self.encoder_projector.forward_hip(enc, pmask, stash)
        else:
            with trace.phase("projector"):
                proj = self.encoder_projector.forward_hip(enc, stash)  # [B, Ta, dl]
        Ta = proj.shape[1]
        if modality_mask is None:
            raise RuntimeError("modality_mask is required (speech recipes always provide it)")
        mm = modality_mask.to(torch.uint8).contiguous()
        embeds, spans = ops.embed_splice_fwd(input_ids, mm, self.llm.embed, proj)
        if kwargs.get("inference_mode", False):
            return embeds.view(B, T, -1), attention_mask
        targets, n_valid = early_targets if early_targets is not None else (None, None)
        if labels is not None and targets is None:
            targets, n_valid = ops.ce_targets(labels.contiguous())
        want_logits = self.return_logits if self.return_logits is not None else (not train)
        pack_idx = None
        if self.cfg.get("varlen", False):
            am = attention_mask.to(torch.bool)
            # packing is exact only for RIGHT-padded batches (MultiTaskDataset collator): every sequence then starts at
            # position 0 exactly like its padded row; left-padded rows keep the padded path (RoPE offset quirk)
            if bool((am[:, 1:] <= am[:, :-1]).all()) and bool(am[:, 0].all()):
                pack_idx = am.flatten().nonzero().squeeze(1)
        if pack_idx is not None:
            # ragged batch without pad tokens: one packed sequence of sum(len) rows; attention is restricted to each
            # sequence by seg_lo / seg_hi, rotary positions restart at every sequence
            lens = am.sum(1)
            Mp = int(pack_idx.numel())
            starts = lens.cumsum(0) - lens
            lo = starts.repeat_interleave(lens, output_size=Mp).to(torch.int32).contiguous()
            hi = (starts + lens).repeat_interleave(lens, output_size=Mp).to(torch.int32).contiguous()
            pos = (torch.arange(Mp, device=dev, dtype=torch.int32) - lo).contiguous()
            pack_idx32 = pack_idx.to(torch.int32)
            h_packed = ops.gather_rows(embeds, pack_idx32)
            t_packed = targets.index_select(0, pack_idx).contiguous() if targets is not None else None
            with trace.phase("llm_fwd"):
                out2, logits_p, lstash = self.llm.forward_hip(h_packed, 1, Mp, None, t_packed, n_valid, train, want_logits,
                                                              packed=(pos, lo, hi, T), label_count=label_count)
            logits = None
            if logits_p is not None:   # back to the padded [B*T, V] layout (pad rows zero)
                logits = torch.zeros((B * T, logits_p.shape[1]), dtype=logits_p.dtype, device=dev)
                logits.index_copy_(0, pack_idx, logits_p)
        else:
            Tp = round_up(T, 64)
            key_mask = torch.zeros((B, Tp), dtype=torch.uint8, device=dev)
            key_mask[:, :T] = attention_mask.to(torch.uint8)
            with trace.phase("llm_fwd"):
                out2, logits, lstash = self.llm.forward_hip(embeds, B, T, key_mask, targets, n_valid, train, want_logits, label_count=label_count)
        loss = acc = None
        if out2 is not None:
            loss_val, acc = out2[0], out2[1]
            if train:
                stash.update(lstash)
                stash.update(spans=spans, Ta=Ta, batch_B=B, batch_T=T, pack_idx=pack_idx)  # (B, T of the LLM pass itself live in lstash)
                plist = tuple(self.store.params.values()) if self.autograd_params else ()
                loss = _SlamStep.apply(self._anchor, self, stash, loss_val, *plist)
            else:
                loss = loss_val
        if not self.metric:
            acc = -1
        outputs = SimpleNamespace(loss=loss, logits=logits.view(B, T, -1) if logits is not None else None)
        return outputs, acc

    # ---- ragged encoder (++model_config.varlen_encoder=true) ---------------------------------------------
    def _ragged_frames(self, audio_mel: torch.Tensor, kwargs) -> Optional[List[int]]:
        """real mel frames per clip as host ints, or None when the batch is not ragged (equal lengths, or pad_or_trim on:
        every clip is 3000 frames) -- then the reference-padded path runs and results equal the reference's bit for bit.
        `audio_len_list` (a python list the collators add; the reference's train loop moves tensors only,
        utils/train_utils.py:101-111) avoids the device -> host sync that `audio_len.tolist()` costs."""
        if self.cfg.get("pad_or_trim", True):
            return None
        lens = kwargs.get("audio_len_list", None)
        if lens is None:
            al = kwargs.get("audio_len", None)
            if al is None:
                return None
            lens = al.tolist()
        frames = [min(int(n), 480000) // 160 for n in lens]
        if len(frames) != audio_mel.shape[0] or min(frames) < 1:
            return None
        if min(frames) == max(frames) == audio_mel.shape[1]:
            return None
        return frames

    def _encode_project_ragged(self, audio_mel: torch.Tensor, n_frames: List[int], stash: Optional[dict]) -> torch.Tensor:
        """packed encoder -> projector -> padded [B, Ta_max, dl] for the splice (rows past a clip's own audio tokens are zero,
        which is what the splice's clamp `min(sum(mask), Ta)` never reads anyway)."""
        dev = self.device_
        enc_train = self.train_encoder and stash is not None      # round 5: un-frozen Whisper on the ragged layout
        if enc_train:
            enc, T2 = self.encoder.forward_packed_train(audio_mel, n_frames, stash)
        else:
            enc, T2 = self.encoder.forward_packed(audio_mel, n_frames)
        B, d = len(T2), enc.shape[1]
        if self.projector_name == "q-former":   # cross-attends over the frames under a key mask: hand it the padded layout
            T2max = max(T2)
            inv = torch.full((B, T2max), -1, dtype=torch.int32)
            acc = 0
            for b_, t2 in enumerate(T2):
                inv[b_, :t2] = torch.arange(acc, acc + t2, dtype=torch.int32)
                acc += t2
            if enc_train:      # dL/d(padded layout) -> packed rows: row i of the packed output sits at padded row enc_rows[i]
                stash["ragged_enc_rows"] = torch.nonzero(inv.view(-1) >= 0).view(-1).to(torch.int32).to(dev, non_blocking=True)
            inv = inv.to(dev, non_blocking=True)
            encp = ops.gather_rows(enc, inv.view(-1)).view(B, T2max, d)
            return self.encoder_projector.forward_hip(encp, (inv >= 0).to(torch.float32), stash)
        k = self.cfg["ds_rate"]
        Ta = [t2 // k for t2 in T2]
        Tam = max(Ta)
        if Tam < 1:
            raise RuntimeError("every clip of the batch is shorter than one projector frame")
        win, inv, valid = [], torch.full((B, Tam), -1, dtype=torch.int32), []
        acc = row = 0
        for b_, (t2, ta) in enumerate(zip(T2, Ta)):
            win.append(acc + k * torch.arange(ta, dtype=torch.int32))      # first encoder row of every k-frame window
            inv[b_, :ta] = torch.arange(row, row + ta, dtype=torch.int32)
            valid.append(b_ * Tam + torch.arange(ta, dtype=torch.int32))
            acc += t2
            row += ta
        if enc_train:
            # adjoint of the window gather below: packed encoder row r = win[i] + j is row i * k + j of dL/d(xp) viewed as [sum Ta * k, d];
            # the t2 % k last frames of a clip belong to no window (the reference's view(B, T // k, k * d) drops them): zero gradient
            unwin = torch.full((sum(T2),), -1, dtype=torch.int32)
            w0 = torch.cat(win).to(torch.int64)                               # one scatter for all windows (was a slice assignment per window)
            unwin[(w0[:, None] + torch.arange(k)).flatten()] = torch.arange(w0.numel() * k, dtype=torch.int32)
            stash["ragged_unwindow"] = unwin.to(dev, non_blocking=True)
        win = torch.cat(win).to(dev, non_blocking=True)
        inv = inv.view(-1).to(dev, non_blocking=True)
        xp = ops.gather_rows(enc, win, width=k * d)                        # [sum Ta, k*d]: the reference's view(B, T//k, k*d) per clip
        y = self.encoder_projector.forward_rows(xp, stash)                 # [sum Ta, dl]
        if stash is not None:
            stash["proj_valid_rows"] = torch.cat(valid).to(dev, non_blocking=True)
        return ops.gather_rows(y, inv).view(B, Tam, y.shape[1])

    def _run_backward(self, stash: dict, grad_out: torch.Tensor, as_autograd: bool = False):
        st = self.store
        if as_autograd:
            if self.grad_hooks:
                raise RuntimeError("autograd_params mode hands gradients to autograd/DDP; detach GradSync")
            # fresh buffer: what we return belongs to autograd (adopted as .grad or added into it) and must not be
            # overwritten by the next backward
            # (zero-filled: the alignment gaps between parameters whose size is not a multiple of 64 are never written by a kernel,
            # and the fused optimizers sweep the whole flat buffer -- uninitialised gap values must not reach the moments)
            st.grad = torch.zeros_like(st.flat)
            accumulate = False
        else:
            accumulate = any(p.grad is not None for p in st.params.values())
            if accumulate and st.pure_bf16:
                # bf16 masters: `.grad` is the bf16 buffer -- whatever the loop did to it since the last backward (unscale, clip, zero_)
                # is what the next micro-step adds to, like autograd's bf16 accumulation in the reference
                st.grad.copy_(st.grad_lp)
        for hk in self.grad_hooks:
            begin = getattr(hk, "on_backward_begin", None)
            if begin is not None:
                begin()
        gs = grad_out.reshape(1).to(torch.float32).contiguous()
        with trace.phase("llm_bwd"):
            dh0 = self.llm.backward_hip(stash, gs, accumulate, on_layer_done=self._on_layer_done)
        if stash.get("pack_idx") is not None:   # packed rows -> padded [B*T, d] layout (pad rows carry no gradient)
            n_rows = stash["batch_B"] * stash["batch_T"]
            inv = torch.full((n_rows,), -1, dtype=torch.int32, device=dh0.device)
            inv[stash["pack_idx"]] = torch.arange(dh0.shape[0], dtype=torch.int32, device=dh0.device)
            dh0 = ops.gather_rows(dh0, inv)   # pad rows come out zero
        dproj = ops.embed_splice_bwd(stash["spans"], dh0, stash["batch_B"], stash["batch_T"], stash["Ta"], self.cfg["llm_dim"])
        if stash.get("proj_valid_rows") is not None:   # ragged encoder: the projector ran on the clips' own rows only
            dproj = ops.gather_rows(dproj, stash["proj_valid_rows"])
        with trace.phase("projector_bwd"):
            d_enc = self.encoder_projector.backward_hip(dproj, stash, accumulate)
        if self.train_encoder:
            if stash.get("ragged_unwindow") is not None:      # ragged encoder, stacked-row projectors: dL/d(k-frame windows) -> packed encoder rows
                d_enc = ops.gather_rows(d_enc.reshape(-1, self.cfg["enc_dim"]), stash.pop("ragged_unwindow"))
            elif stash.get("ragged_enc_rows") is not None:    # ragged encoder, Q-Former: padded [B * T2max, d] -> packed rows
                d_enc = ops.gather_rows(d_enc.reshape(-1, self.cfg["enc_dim"]), stash.pop("ragged_enc_rows"))
            with trace.phase("encoder_bwd"):
                self.encoder.backward_hip(d_enc, stash, accumulate)
        if st.pure_bf16:       # gradients leave in the parameters' dtype: one rounding of the fp32 sums per backward
            if as_autograd:
                fresh = torch.empty_like(st.grad_lp)
                ops.cast_bf16(st.grad, fresh)
                return [fresh[off:off + n].view(shape) for (off, n, shape) in (st.offsets[name] for name in st.params)]
            ops.cast_bf16(st.grad, st.grad_lp)
        if as_autograd:
            return [st.grad_view(name) for name in st.params]
        for name, p in st.params.items():
            if p.grad is None:
                p.grad = st.grad_lp_view(name) if st.pure_bf16 else st.grad_view(name)
        for hk in self.grad_hooks:
            hk.on_prefix(st.size) if hasattr(hk, "on_prefix") else hk(st.size)
        return ()

    def layer_prefix_ends(self) -> Dict[int, int]:
        """{LLM layer index: end offset of its LoRA gradients in the flat buffer} (reserved last layer first: a prefix per layer)"""
        ends = getattr(self, "_layer_prefix_end", None)
        if ends is None:
            ends = {}
            for n, (off, cnt, _) in self.store.offsets.items():
                if n.startswith("llm.") and ".layers." in n:    # (a trainable HuBERT / WavLM encoder has `.layers.` names too, far behind the prefix)
                    k = int(n.split(".layers.")[1].split(".")[0])
                    ends[k] = max(ends.get(k, 0), off + round_up(cnt, 64))
            self._layer_prefix_end = ends
        return ends

    def prefix_plan(self) -> List[int]:
        """the sequence of `on_prefix(end)` announcements one backward makes (layers last to first, then the whole buffer): what an
        exhausted rank replays over a zero buffer (train.GradSync.shadow_backward, the `Join` policy)"""
        ends = self.layer_prefix_ends()
        return [ends[li] for li in reversed(range(len(self.llm.layers))) if li in ends] + [self.store.size]

    def attach_grad_views(self):
        """`.grad` of every trainable parameter = its view of the flat gradient buffer (what a backward leaves behind)"""
        st = self.store
        for name, p in st.params.items():
            if p.grad is None:
                p.grad = st.grad_lp_view(name) if st.pure_bf16 else st.grad_view(name)

    def _on_layer_done(self, li: int):
        if not self.grad_hooks:
            return
        # LoRA grads of layers >= li are final: they occupy the prefix of the flat buffer (reserved last layer first)
        ends = self.layer_prefix_ends()
        end = ends.get(li)
        if end is None:
            return
        for hk in self.grad_hooks:
            hk.on_prefix(end) if hasattr(hk, "on_prefix") else hk(end)

    # ---- generate ------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, **kwargs):
        """slam_model.generate (src/slam_llm/models/slam_model.py:409-456): forward(..., inference_mode=True) for
        the spliced prompt embeddings, then HF `llm.generate(inputs_embeds=..., attention_mask=..., num_beams=4,
        max_new_tokens=200, min_length=1, length_penalty=1.0, eos/pad from the tokenizer)`.  Here: one prefill
        pass + KV-cache decode steps on the HIP path; beam / greedy / sampling bookkeeping in slam_llm_amd/decode.py
        (do_sample=True: temperature / top_k / top_p warpers + multinomial draws, greedy or beam-sample like HF).
        Returns the NEW tokens only [B, <=max_new_tokens] (HF's behaviour for inputs_embeds prompts)."""
        from . import decode
        sample = None
        if kwargs.get("do_sample", False):
            # HF's GenerationConfig default top_k is 50 in the transformers 4.x line the reference was written against (unset in
            # 5.x); top_k=0 / None disables it.  `generator=` (optional) makes the draws reproducible.
            tk = kwargs.get("top_k", 50)
            sample = dict(temperature=float(kwargs.get("temperature", 1.0)), top_k=int(tk) if tk else 0,
                          top_p=float(kwargs.get("top_p", 1.0)), generator=kwargs.get("generator"))
            if sample["temperature"] <= 0.0 or not 0.0 < sample["top_p"] <= 1.0:
                raise ValueError(f"generate: temperature must be > 0 and top_p in (0, 1] (got {sample['temperature']}, {sample['top_p']})")
        tok = self.tokenizer
        eos = kwargs.get("eos_token_id", getattr(tok, "eos_token_id", None))
        pad = kwargs.get("pad_token_id", getattr(tok, "pad_token_id", None))
        if eos is None or pad is None:
            raise RuntimeError("generate() needs eos/pad token ids (tokenizer or eos_token_id= / pad_token_id=)")
        max_new = int(kwargs.get("max_new_tokens", 200))
        num_beams = int(kwargs.get("num_beams", 4))
        fwd_kwargs = {k_: v for k_, v in kwargs.items() if k_ not in decode.GENERATE_KEYS}
        fwd_kwargs["inference_mode"] = True
        embeds, attention_mask = self.forward(input_ids=input_ids, attention_mask=attention_mask, **fwd_kwargs)
        B, T, d = embeds.shape
        logits0, cache = self.llm.prefill(embeds.reshape(B * T, d), B, T, attention_mask, max_new, beams=num_beams)
        state = {"first": True}

        def step_fn(tokens, src_rows):
            if state["first"]:       # prompt logits, one row per hypothesis (the prompt KV itself is shared)
                state["first"] = False
                return logits0 if num_beams == 1 else logits0.repeat_interleave(num_beams, dim=0)
            if src_rows is not None:  # beam search: hypothesis r continues old hypothesis src_rows[r]
                self.llm.reorder_cache(cache, src_rows)
            return self.llm.decode_step(tokens[:, -1], cache)

        if num_beams == 1:
            return decode.greedy_search(step_fn, B, max_new, int(eos), int(pad), int(kwargs.get("min_length", 1)),
                                        self.device_, float(kwargs.get("repetition_penalty", 1.0)), sample)
        return decode.beam_search(step_fn, B, num_beams, max_new, int(eos), int(pad), int(kwargs.get("min_length", 1)),
                                  float(kwargs.get("length_penalty", 1.0)), self.device_,
                                  float(kwargs.get("repetition_penalty", 1.0)), sample)


    @torch.no_grad()
    def inference(self, wav_path=None, prompt=None, **kwargs):
        """single-utterance decode, examples/asr_librispeech/model/slam_model_asr.py:81-152: load + pad_or_trim the
        wav, [audio, "USER: {prompt}\n ASSISTANT:"] prompt, generate(**kwargs).  The log-mel runs on the device."""
        import os
        from .batcher import collate, make_sample, whisper_audio_length
        from .dataset import load_wav_16k
        if not wav_path or not os.path.exists(wav_path):
            raise NotImplementedError("text-only QA (no audio) is not part of the speech hot path")
        audio = load_wav_16k(wav_path)
        ids = self.tokenizer.encode("USER: {}\n ASSISTANT:".format(prompt))
        alen = whisper_audio_length(len(audio), self.cfg["ds_rate"])
        batch = collate([make_sample(audio, ids, None, self.tokenizer.eos_token_id, alen)], self.tokenizer.pad_token_id, True)
        batch = {k: (v.to(self.device_) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        return self.generate(**batch, **kwargs)


# ======================================================================================== optimizer
class SlamAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (src/slam_llm/pipeline/finetune.py:247-251) as ONE fused kernel over the
    flat master/grad buffers; also refreshes the bf16 compute copies.  Works with LambdaLR (reads group['lr']).
    The moments and the step counter round-trip through state_dict()/load_state_dict() (keys `exp_avg`, `exp_avg_sq`
    as flat tensors in the store's layout, `step`), so a resumed run continues the bias correction."""

    def __init__(self, model: SlamHipModel, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = list(model.store.params.values())
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        st = model.store
        if st.pure_bf16 and type(self) is SlamAdamW:
            raise NotImplementedError("SlamAdamW keeps fp32 masters; after model.to(torch.bfloat16) (pure_bf16 route) use torch.optim.AdamW(model.parameters()) "
                                      "or SlamAnyPrecisionAdamW(model, pure_bf16=True) like the reference does (pipeline/finetune.py:237-251)")
        self.exp_avg = torch.zeros_like(st.flat)
        self.exp_avg_sq = torch.zeros_like(st.flat)
        self._step = 0

    def _flat_grad(self) -> torch.Tensor:
        """the flat gradient buffer the kernel reads.  Flat-buffer mode: `.grad` of every parameter already views it.
        autograd_params mode: autograd owns `.grad` (views of the last backward's buffer when it adopted them, other
        tensors after accumulation or under DDP's bucket views) -> gather whatever is not already in place."""
        st = self.model.store
        if st.pure_bf16:      # bf16 masters: the loop's gradients are the bf16 buffer (what a bf16 model's autograd would hold)
            lp = st.grad_lp.data_ptr()
            if all(p.grad is not None and p.grad.data_ptr() == lp + 2 * st.offsets[n][0] for n, p in st.params.items()):
                st.grad.copy_(st.grad_lp)
                return st.grad
        base = st.grad.data_ptr()
        for name, p in st.params.items():
            g = p.grad
            if g is None:
                raise RuntimeError(f"SlamAdamW.step(): parameter {name} has no gradient (all trainables are produced by every backward)")
            off = st.offsets[name][0]
            if g.data_ptr() != base + 4 * off:
                st.grad_view(name).copy_(g)
        return st.grad

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        st = self.model.store
        self._step += 1
        with trace.phase("optimizer"):
            ops.adamw_step(st.flat, self._flat_grad(), self.exp_avg, self.exp_avg_sq, st.flat_bf16, float(g["lr"]), g["betas"][0],
                           g["betas"][1], g["eps"], g["weight_decay"], self._step)
            self.model.refresh_derived()
        self.model._always_refresh = False

    def zero_grad(self, set_to_none: bool = True):
        for p in self.model.store.params.values():
            p.grad = None

    def state_dict(self):
        sd = super().state_dict()
        sd["slam"] = dict(exp_avg=self.exp_avg.detach().cpu(), exp_avg_sq=self.exp_avg_sq.detach().cpu(), step=self._step,
                          layout=[(n, list(shape), off) for n, shape, off in self.model.store.entries])
        return sd

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        slam = sd.pop("slam", None)
        super().load_state_dict(sd)
        if slam is None:
            raise KeyError("SlamAdamW.load_state_dict: no 'slam' entry (moments / step): not a SlamAdamW checkpoint")
        layout = [(n, list(shape), off) for n, shape, off in self.model.store.entries]
        if [tuple(map(str, e)) for e in slam["layout"]] != [tuple(map(str, e)) for e in layout]:
            raise ValueError("SlamAdamW.load_state_dict: the checkpoint's parameter layout differs from this model's")
        self.exp_avg.copy_(slam["exp_avg"])
        self.exp_avg_sq.copy_(slam["exp_avg_sq"])
        self._step = int(slam["step"])


class SlamAnyPrecisionAdamW(SlamAdamW):
    """AnyPrecisionAdamW (src/slam_llm/policies/anyprecision_optimizer.py:16-178) as one fused kernel over the flat buffers: bf16
    momentum and variance as pipeline/finetune.py:237-245 configures it, optional bf16 Kahan compensation.  `pure_bf16=True`
    reproduces the reference's `model.to(torch.bfloat16)` route (finetune.py:154-155): the fp32 master buffer then only
    ever holds bf16-representable values (the parameters are rounded once here) and every parameter update rounds like a bf16
    tensor op would; False keeps this build's fp32 masters under bf16 states."""

    def __init__(self, model: SlamHipModel, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, use_kahan_summation=False,
                 pure_bf16=False):
        super().__init__(model, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        st = model.store
        self.exp_avg = torch.zeros_like(st.flat, dtype=torch.bfloat16)
        self.exp_avg_sq = torch.zeros_like(st.flat, dtype=torch.bfloat16)
        self.compensation = torch.zeros_like(st.flat, dtype=torch.bfloat16) if use_kahan_summation else None
        self.pure_bf16 = bool(pure_bf16) or st.pure_bf16      # (after model.to(torch.bfloat16) the parameters ARE bf16)
        if self.pure_bf16 and not st.pure_bf16:
            with torch.no_grad():
                st.flat.copy_(st.flat.to(torch.bfloat16).float())
            model.mark_params_updated()

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        st = self.model.store
        self._step += 1
        ops.adamw_anyprecision_step(st.flat, self._flat_grad(), self.exp_avg, self.exp_avg_sq, self.compensation, st.flat_bf16,
                                    float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self._step,
                                    params_are_bf16=self.pure_bf16)
        self.model.refresh_derived()
        self.model._always_refresh = False

    def state_dict(self):
        sd = super().state_dict()
        if self.compensation is not None:
            sd["slam"]["compensation"] = self.compensation.detach().cpu()
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        if self.compensation is not None:
            self.compensation.copy_(state_dict["slam"]["compensation"])
