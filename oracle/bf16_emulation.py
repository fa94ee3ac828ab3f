"""bf16-EMULATING twin of oracle/slam_oracle.py:llama_forward -- TEST INFRASTRUCTURE ONLY (same import rule as slam_oracle.py).

Why it exists (VERDICT r5 next #1b).  The HIP path meets the fp32 oracle at full depth (32 + 32 layers, true widths) with gradient
cosines >= 0.9996 on the v_proj adapters and the projector, but the q_proj adapters fall with depth to ~0.994.  The explanation given
in round 5 -- dQ = sum_k dS_k K_k with sum_k dS_k = 0 is a cancelling sum, so the bf16 rounding of the operands the path keeps in
bf16 is amplified where the keys share a large common component -- was prose.  This file turns it into an experiment: the SAME fp32
arithmetic as the oracle, with a rounding to bf16 inserted at every tensor the HIP path materialises in bf16 (forward AND backward),
each site switchable, so that
  * emulated-vs-fp32 must reproduce the degradation of the HIP path tensor by tensor if rounding is its cause, and
  * switching single sites off shows WHICH rounding carries it.
Nothing here restates the reference (the reference computes in fp32 / fp16 autocast): it restates the HIP path's number formats over
the reference's arithmetic.  Sites, by the name used in `sites`:

  weights   frozen W are bf16 operands (model.py:FusedLinear.finalize); a caller that has rounded W in place leaves this site out
  lora_w    LoRA A and (alpha/r) B are bf16 operands, their gradients stay fp32 (model.py:FusedLinear.refresh, lora_pack_b)
  h         residual stream [M, d] after o_proj + residual and after down_proj + residual; dL/dh likewise
  x         RMSNorm output (HF double rounding: x * rstd -> bf16, * w -> bf16); dL/dx rounded once
  u         LoRA first hop u = x A^T [M, r]; dL/du
  qkv       fused q|k|v product before RoPE (forward only: the backward has no tensor between attention and RoPE)
  rope      q, k after RoPE (in place); dL/d(q, k) after the RoPE backward fused into the attention backward's epilogue
  p         softmax numerators exp(s - m) as the bf16 operand of P V (forward) and the normalised P as the operand of dV = P^T dO
  o         attention output O; dL/dO
  ds        dL/dscore = P (dP - Delta) scale as the bf16 operand of dQ = dS K and dK = dS^T Q
  delta     Delta = sum_d dO O taken from the ROUNDED O (what the kernels do) instead of sum_k P dP
  dv        dL/dV rounded on the way out (dQ / dK leave through `rope`)
  gu        gate | up product; its gradient
  hh        silu(gate) * up; its gradient
  hn        final RMSNorm output; its gradient
  logits    lm_head product in bf16; dL/dlogits written in bf16 over it

`reorder=True` evaluates every matrix product as two half-contractions added last -- the same fp32 products summed in another order,
like any second implementation with a different tiling would.  Two emulations that differ ONLY in that decorrelate over depth (a bf16 rounding turns a 1e-7 difference
into a whole-ulp difference whenever the value sits within 1e-7 of a rounding boundary, and those flips cascade); their distance is the
yardstick for how close two correct bf16 implementations can be expected to be.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import slam_oracle as O

ALL_SITES = frozenset("weights lora_w h x u qkv rope p o ds delta dv gu hh hn logits".split())


def rb(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


class _Rnd(torch.autograd.Function):
    """y = bf16(x); dL/dx = bf16(dL/dy): a tensor the HIP path materialises in bf16 in both passes"""

    @staticmethod
    def forward(ctx, x):
        return rb(x)

    @staticmethod
    def backward(ctx, g):
        return rb(g)


class _RndF(torch.autograd.Function):
    """y = bf16(x) in the forward only"""

    @staticmethod
    def forward(ctx, x):
        return rb(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _Emu:
    def __init__(self, sites, reorder):
        self.sites, self.reorder = frozenset(sites), reorder

    def on(self, s):
        return s in self.sites

    def r(self, s, t):      # both directions
        return _Rnd.apply(t) if s in self.sites else t

    def rf(self, s, t):
        return _RndF.apply(t) if s in self.sites else t

    def mm(self, a, b):
        """a @ b; under `reorder` the same fp32 products summed in another order (the contraction cut in two halves that are added
        last) -- what a second correct implementation with a different tiling does"""
        if self.reorder and a.shape[-1] >= 2:
            h = a.shape[-1] // 2
            return a[..., :h] @ b[..., :h, :] + a[..., h:] @ b[..., h:, :]
        return a @ b

    def lin(self, x, w):
        return self.mm(x, w.t())


class _Attn(torch.autograd.Function):
    """causal GQA attention with the HIP kernels' number formats (csrc/attention.hip): S = Q K^T in fp32 from bf16 operands, online-softmax
    numerators rounded to bf16 for P V, O = acc / l rounded on the way out (by the caller's `o` site), LSE in fp32; backward recomputes P
    from LSE, Delta = sum_d dO O from the rounded O, dS = P (dP - Delta) scale rounded to bf16 for the dQ / dK products, P rounded to
    bf16 for dV."""

    @staticmethod
    def forward(ctx, q, k, v, add_mask, scale, emu):
        # q [B, Hq, T, D], k / v [B, Hq, T, D] (GQA already expanded by the caller: autograd sums the group)
        s = emu.mm(q, k.transpose(2, 3)) * scale + add_mask
        m = s.amax(-1, keepdim=True)
        e = torch.exp(s - m)
        l = e.sum(-1, keepdim=True)                       # fp32 row sums of the UNROUNDED numerators (the kernels sum before the pack)
        eb = rb(e) if emu.on("p") else e
        o = emu.mm(eb, v) / l
        lse = m + torch.log(l)
        ctx.emu, ctx.scale = emu, scale
        ctx.save_for_backward(q, k, v, add_mask, lse)
        ctx.o_holder = {}
        return o, lse

    @staticmethod
    def backward(ctx, do, _dlse):
        q, k, v, add_mask, lse = ctx.saved_tensors
        emu, scale = ctx.emu, ctx.scale
        o = ctx.o_holder["o"]                             # what the forward's consumer saw (rounded under `o`)
        s = emu.mm(q, k.transpose(2, 3)) * scale + add_mask
        p = torch.exp(s - lse)
        dp = emu.mm(do, v.transpose(2, 3))
        if emu.on("delta"):
            delta = (do * o).sum(-1, keepdim=True)
        else:
            delta = (p * dp).sum(-1, keepdim=True)
        ds = p * (dp - delta) * scale
        dsb = rb(ds) if emu.on("ds") else ds
        pb = rb(p) if emu.on("p") else p
        dq = emu.mm(dsb, k)
        dk = emu.mm(dsb.transpose(2, 3), q)
        dv = emu.mm(pb.transpose(2, 3), do)
        if emu.on("dv"):
            dv = rb(dv)
        return dq, dk, dv, None, None, None


def llama_forward_emulated(W, cfg, inputs_embeds, attention_mask, labels=None, prefix="llm.base_model.model.",
                           sites=ALL_SITES, reorder=False, capture=None):
    """oracle.slam_oracle.llama_forward (same arguments, same returns) under the HIP path's number formats -- see the module docstring.
    `capture`: optional dict that receives {layer index: dict(q, k, v, o, lse)} of the layers listed in capture['layers'] (tensors are
    detached clones in the [B, H, T, D] layout) for the teacher-forced kernel checks."""
    emu = _Emu(sites, reorder)
    B, T, d = inputs_embeds.shape
    Hq, Hkv, D = cfg["llm_heads"], cfg["llm_kv_heads"], cfg["llm_head_dim"]
    eps = cfg["rms_eps"]
    sc = cfg["lora_alpha"] / cfg["lora_r"]
    cos, sin = O.rope_tables(T, D, cfg["rope_theta"])
    minv = torch.finfo(torch.float32).min
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
    allowed = causal[None, None] & attention_mask.bool()[:, None, None, :]
    add_mask = torch.zeros(B, 1, T, T).masked_fill(~allowed, minv)
    wq = (lambda t: rb(t)) if emu.on("weights") else (lambda t: t)

    def rms(x, w):
        v = x.pow(2).mean(-1, keepdim=True)
        xh = emu.rf("x", x * torch.rsqrt(v + eps))       # HF: (x * rstd).to(dtype), then weight * that
        return emu.r("x", w * xh)

    def lora_lin(name, x, res=None, site="h"):
        y = emu.lin(x, wq(W[name + ".weight"]))
        a = W.get(name + ".lora_A.default.weight")
        if a is not None:
            b = W[name + ".lora_B.default.weight"]
            u = emu.r("u", emu.lin(x, emu.rf("lora_w", a)))
            y = y + emu.lin(u, emu.rf("lora_w", b * sc))
        if res is not None:
            y = y + res
        return emu.r(site, y)

    h = emu.r("h", inputs_embeds)
    for i in range(cfg["llm_layers"]):
        p = f"{prefix}model.layers.{i}."
        x = rms(h, W[p + "input_layernorm.weight"])
        q = lora_lin(p + "self_attn.q_proj", x, site="none")
        k = lora_lin(p + "self_attn.k_proj", x, site="none")
        v = lora_lin(p + "self_attn.v_proj", x, site="none")
        q, k = emu.rf("qkv", q), emu.rf("qkv", k)
        v = emu.r("qkv", v) if emu.on("qkv") else v       # v has no RoPE: its one tensor is rounded in both passes (dv site rounds the gradient)
        q = q.view(B, T, Hq, D).transpose(1, 2)
        k = k.view(B, T, Hkv, D).transpose(1, 2)
        v = v.view(B, T, Hkv, D).transpose(1, 2)
        q = emu.r("rope", q * cos + O._rot_half(q) * sin)
        k = emu.r("rope", k * cos + O._rot_half(k) * sin)
        rep = Hq // Hkv
        ke = k[:, :, None].expand(B, Hkv, rep, T, D).reshape(B, Hq, T, D)
        ve = v[:, :, None].expand(B, Hkv, rep, T, D).reshape(B, Hq, T, D)
        o, lse = _Attn.apply(q, ke, ve, add_mask, D ** -0.5, emu)
        fn = o.grad_fn
        o = emu.r("o", o)
        if fn is not None:
            fn.o_holder["o"] = o.detach()
        if capture is not None and i in capture.get("layers", ()):
            capture[i] = dict(q=q.detach().clone(), k=k.detach().clone(), v=v.detach().clone(), o=o.detach().clone(), lse=lse.detach().clone())
        o = o.transpose(1, 2).reshape(B, T, Hq * D)
        h = lora_lin(p + "self_attn.o_proj", o, res=h, site="h")
        x = rms(h, W[p + "post_attention_layernorm.weight"])
        g = lora_lin(p + "mlp.gate_proj", x, site="gu")
        u_ = lora_lin(p + "mlp.up_proj", x, site="gu")
        hh = emu.r("hh", F.silu(g) * u_)
        h = lora_lin(p + "mlp.down_proj", hh, res=h, site="h")
    v_ = h.pow(2).mean(-1, keepdim=True)
    hn = emu.r("hn", W[prefix + "model.norm.weight"] * emu.rf("hn", h * torch.rsqrt(v_ + eps)))
    logits = emu.r("logits", emu.lin(hn, wq(W[prefix + "lm_head.weight"])))
    loss = None
    if labels is not None:
        sl = F.pad(labels, (0, 1), value=-100)[..., 1:].contiguous()
        loss = F.cross_entropy(logits.view(-1, logits.shape[-1]), sl.view(-1), ignore_index=-100, reduction="mean")
    return loss, logits


def projector_concat_emulated(W, x: torch.Tensor, k: int, prefix="encoder_projector.", sites=ALL_SITES, reorder=False) -> torch.Tensor:
    """oracle.slam_oracle.projector_concat under the HIP path's formats: bf16 encoder output, bf16 compute copies of the projector's
    weights (biases are read in fp32), bf16 hidden and output activations (model.py:HipProjectorConcat)."""
    emu = _Emu(sites, reorder)
    B, T, d = x.shape
    drop = T % k
    if drop > 0:
        x = x[:, :-drop, :]
    x = emu.r("h", x.contiguous().view(B, x.shape[1] // k, d * k))
    # (the projector's weights are TRAINABLE: fp32 masters with a bf16 compute copy, like the adapters -- the `lora_w` site)
    h1 = emu.r("h", F.relu(emu.lin(x, emu.rf("lora_w", W[prefix + "linear1.weight"])) + W[prefix + "linear1.bias"]))
    return emu.r("h", emu.lin(h1, emu.rf("lora_w", W[prefix + "linear2.weight"])) + W[prefix + "linear2.bias"])
