"""Q-Former projector on the HIP path (trainable, forward + hand-written backward).

Mirrors `EncoderProjectorQFormer` (src/slam_llm/models/projector.py:51-80): learned queries [1, Q, 768] ->
HF `Blip2QFormerModel` (BERT-style post-LN layers: self-attention over the queries, cross-attention to the encoder
states every 2nd layer with `encoder_attention_mask`, GELU feed-forward; transformers/models/blip_2/
modeling_blip_2.py:536-760, 849-940) -> Linear(768 -> llm_dim) -> LayerNorm(llm_dim, 1e-5).
State-dict keys are the reference's (`encoder_projector.query`, `encoder_projector.qformer.encoder.layer.N...`).
Dropout: the reference's Blip2QFormerConfig() defaults (hidden_dropout_prob = attention_probs_dropout_prob = 0.1) are
live in train mode.  The four HIDDEN dropouts of the stack (after the query LayerNorm, and after each of the self-attention /
cross-attention / feed-forward output projections, before the residual add -- modeling_blip_2.py Blip2QFormerSelfOutput /
Blip2QFormerOutput / Blip2QFormerModel.forward) are applied with the counter-based mask of slam_dropout_bf16 (recomputed,
never stored, in the backward; `qf_dropout`, default 0.1, 0 in eval mode).  The dropout on the attention PROBABILITIES
(`attention_probs_dropout_prob`, Blip2QFormerMultiHeadAttention) is applied inside the fused attention kernels (same mask
generator, one seed per attention call, recomputed by the backward kernels).

All products are NT GEMMs (weights' transposes are refreshed once per optimizer step), attention runs on the MFMA
kernels (self: Tq = Tk = Q; cross: Tq = Q, Tk = encoder frames with the key-padding mask), weight gradients are
(bf16 transpose -> NT GEMM with fp32 output) written straight into the flat gradient buffer, bias / LayerNorm
gradients are fixed-order column reductions.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .ops import round_up


class HipProjectorQFormer(nn.Module):
    def __init__(self, cfg: dict, store, prefix="encoder_projector."):
        super().__init__()
        self.cfg, self.store, self.prefix = cfg, store, prefix
        self.d, self.H, self.F = cfg.get("qf_dim", 768), cfg.get("qf_heads", 12), cfg.get("qf_ffn", 3072)
        self.L, self.Q = cfg["qf_layers"], cfg["qf_queries"]
        self.eps, self.cross_freq = cfg.get("qf_eps", 1e-12), cfg.get("qf_cross_freq", 2)
        self.d_enc, self.dl = cfg["enc_dim"], cfg["llm_dim"]
        self.p_drop = float(cfg.get("qf_dropout", 0.1) or 0.0)
        self._drop_calls = 0
        d, Fd = self.d, self.F
        assert d % 64 == 0 and d // self.H == 64 and Fd % 64 == 0 and self.d_enc % 64 == 0 and self.dl % 64 == 0
        r, p = store.reserve, prefix
        r(p + "query", (1, self.Q, d))
        P = p + "qformer."
        r(P + "layernorm.weight", (d,)); r(P + "layernorm.bias", (d,))

        def attn(a, kv):
            # query|key|value weights (and biases) are reserved back to back: the bf16 copies form one fused [.., K] operand
            for n, kin in (("query", d), ("key", kv), ("value", kv)):
                r(a + f"attention.{n}.weight", (d, kin))
            for n in ("query", "key", "value"):
                r(a + f"attention.{n}.bias", (d,))
            r(a + "output.dense.weight", (d, d)); r(a + "output.dense.bias", (d,))
            r(a + "output.LayerNorm.weight", (d,)); r(a + "output.LayerNorm.bias", (d,))

        for l in range(self.L):
            Lp = f"{P}encoder.layer.{l}."
            attn(Lp + "attention.", d)
            if l % self.cross_freq == 0:
                attn(Lp + "crossattention.", self.d_enc)
            r(Lp + "intermediate_query.dense.weight", (Fd, d)); r(Lp + "intermediate_query.dense.bias", (Fd,))
            r(Lp + "output_query.dense.weight", (d, Fd)); r(Lp + "output_query.dense.bias", (d,))
            r(Lp + "output_query.LayerNorm.weight", (d,)); r(Lp + "output_query.LayerNorm.bias", (d,))
        r(p + "linear.weight", (self.dl, d)); r(p + "linear.bias", (self.dl,))
        r(p + "norm.weight", (self.dl,)); r(p + "norm.bias", (self.dl,))
        self.wT = {}
        self.need_dx = False   # True when the encoder is trainable: backward_hip then returns dL/d(encoder output)

    # ---- plumbing --------------------------------------------------------------------------------------
    def bind(self):
        from .model import _attach
        for name, prm in self.store.params.items():
            if name.startswith(self.prefix):
                _attach(self, name[len(self.prefix):], prm)

    def _fused(self, first_name: str, rows: int, cols: int, grad=False, master=False):
        """contiguous [rows, cols] view starting at parameter `first_name` (q|k|v packed back to back)"""
        off = self.store.offsets[first_name][0]
        buf = self.store.grad if grad else (self.store.flat if master else self.store.flat_bf16)
        return buf[off: off + rows * cols].view(rows, cols)

    def refresh(self):
        """bf16 transposes of every weight that needs dL/dx (all but the cross K/V projections and nothing else)"""
        st, d = self.store, self.d
        old = self.wT if isinstance(getattr(self, "wT", None), dict) else {}     # updated in place after the first call (ops.transpose_into)
        self.wT = {}
        for name, (off, n, shape) in st.offsets.items():
            if not name.startswith(self.prefix) or not name.endswith(".weight") or len(shape) != 2:
                continue
            if "crossattention.attention.key" in name:
                if self.need_dx:   # trainable encoder: dL/d(encoder states) = d[k | v] . [Wk ; Wv] (fused, reserved back to back)
                    w = self._fused(name, 2 * d, self.d_enc)
                    self.wT[name.replace("key.weight", "kv")] = ops.transpose_into(old.get(name.replace("key.weight", "kv")), w, 2 * d)
                continue
            if "crossattention.attention.value" in name:
                continue  # covered by the fused k|v transpose above (frozen encoder: d(encoder states) is never needed)
            if ".attention.attention.key" in name or ".attention.attention.value" in name:
                continue  # covered by the fused q|k|v transpose below
            if ".attention.attention.query" in name:
                w = self._fused(name, 3 * d, d)
                self.wT[name.replace("query.weight", "qkv")] = ops.transpose_into(old.get(name.replace("query.weight", "qkv")), w, 3 * d)
                continue
            self.wT[name] = ops.transpose_into(old.get(name), st.bf16_view(name), shape[0])

    def _lin_bwd(self, dy, x, w_first: str, b_first: str, N: int, K: int, acc: bool, wT_key: Optional[str]):
        """gradients of y = x W^T + b for a (possibly fused) weight block starting at w_first; returns dx or None"""
        M = dy.shape[0]
        Mp = round_up(M, 64)
        dyT = ops.transpose(dy, Rp=Mp)
        xT = ops.transpose(x, Rp=Mp)
        ops.gemm_nt(dyT, xT, out=self._fused(w_first, N, K, grad=True), accumulate=acc)
        off = self.store.offsets[b_first][0]
        ops.colsum(dy, self.store.grad[off: off + N], accumulate=acc)
        return ops.gemm_nt(dy, self.wT[wT_key]) if wT_key is not None else None

    # ---- forward ---------------------------------------------------------------------------------------
    def forward_hip(self, enc: torch.Tensor, enc_mask: Optional[torch.Tensor], stash: Optional[dict]):
        """enc [B, Tk, d_enc] bf16, enc_mask [B, Tk] (1 = attend) or None -> [B, Q, llm_dim] bf16"""
        st, P, d, H, Fd, Q = self.store, self.prefix + "qformer.", self.d, self.H, self.F, self.Q
        B, Tk, _ = enc.shape
        M = B * Q
        dev = enc.device
        train = stash is not None
        enc2d = enc.reshape(B * Tk, self.d_enc)
        Tkp = round_up(Tk, 64)
        km = None
        if enc_mask is not None:
            km = torch.zeros((B, Tkp), dtype=torch.uint8, device=dev)
            km[:, :Tk] = (enc_mask > 0).to(torch.uint8)
        f32 = st.master_view
        scale = 64 ** -0.5
        use_drop = train and self.training and self.p_drop > 0.0
        seed = (torch.initial_seed() ^ 0x51F0) if use_drop else 0

        def drop_key():
            """(p, seed, offset) of the next hidden dropout of this forward, or None"""
            if not use_drop:
                return None
            self._drop_calls += 1
            return (self.p_drop, seed, self._drop_calls << 40)

        def attn_drop():
            """(p, seed) of the next attention-probability dropout, or None"""
            if not use_drop:
                return None
            self._drop_calls += 1
            return (self.p_drop, (seed * 0x9E3779B1 + self._drop_calls) & (2 ** 63 - 1))

        def out_proj(x, w_name, b_name, residual, key):
            """dense -> dropout -> + residual (Blip2QFormerSelfOutput / Blip2QFormerOutput up to the LayerNorm)"""
            if key is None:
                return ops.gemm_nt(x, st.bf16_view(w_name), bias=f32(b_name), residual=residual)
            t = ops.gemm_nt(x, st.bf16_view(w_name), bias=f32(b_name))
            return ops.dropout(t, *key, out=residual.clone(), accumulate=True)

        q0 = st.bf16_view(self.prefix + "query").view(Q, d)
        h0, m0, r0 = ops.layernorm(q0, f32(P + "layernorm.weight"), f32(P + "layernorm.bias"), self.eps, stats=True)
        h = h0.unsqueeze(0).expand(B, Q, d).reshape(M, d).contiguous()
        k0 = drop_key()
        if k0 is not None:
            h = ops.dropout(h, *k0)
        S = {"layers": [], "B": B, "Tk": Tk, "km": km, "enc2d": enc2d, "q0": q0, "m0": m0, "r0": r0, "k0": k0}
        for l in range(self.L):
            Lp = f"{P}encoder.layer.{l}."
            A = Lp + "attention."
            qkv = ops.gemm_nt(h, self._fused(A + "attention.query.weight", 3 * d, d), bias=self._fused(A + "attention.query.bias", 1, 3 * d, master=True).view(-1))
            ad = attn_drop()
            a, lse = ops.attn_fwd(qkv[:, :d], qkv[:, d: 2 * d], qkv[:, 2 * d:], B, Q, H, H, 64, False, scale, want_lse=train, drop=ad)
            k1 = drop_key()
            s1 = out_proj(a, A + "output.dense.weight", A + "output.dense.bias", h, k1)
            h1, m1, r1 = ops.layernorm(s1, f32(A + "output.LayerNorm.weight"), f32(A + "output.LayerNorm.bias"), self.eps, stats=True)
            rec = dict(h=h, qkv=qkv, a=a, lse=lse, s1=s1, m1=m1, r1=r1, h1=h1, cross=None, k1=k1, ad=ad)
            hx = h1
            if l % self.cross_freq == 0:
                C = Lp + "crossattention."
                qc = ops.gemm_nt(h1, st.bf16_view(C + "attention.query.weight"), bias=f32(C + "attention.query.bias"))
                kvc = ops.gemm_nt(enc2d, self._fused(C + "attention.key.weight", 2 * d, self.d_enc),
                                  bias=self._fused(C + "attention.key.bias", 1, 2 * d, master=True).view(-1))
                adc = attn_drop()
                c, lsec = ops.attn_fwd(qc, kvc[:, :d], kvc[:, d:], B, Q, H, H, 64, False, scale, key_mask=km, want_lse=train, Tk=Tk, drop=adc)
                k2 = drop_key()
                s2 = out_proj(c, C + "output.dense.weight", C + "output.dense.bias", h1, k2)
                hx, mc, rc = ops.layernorm(s2, f32(C + "output.LayerNorm.weight"), f32(C + "output.LayerNorm.bias"), self.eps, stats=True)
                rec["cross"] = dict(qc=qc, kvc=kvc, c=c, lse=lsec, s2=s2, mc=mc, rc=rc, hx=hx, k2=k2, ad=adc)
            z = ops.gemm_nt(hx, st.bf16_view(Lp + "intermediate_query.dense.weight"), bias=f32(Lp + "intermediate_query.dense.bias"))
            f = ops.gelu_fwd(z)
            k3 = drop_key()
            s3 = out_proj(f, Lp + "output_query.dense.weight", Lp + "output_query.dense.bias", hx, k3)
            h, m3, r3 = ops.layernorm(s3, f32(Lp + "output_query.LayerNorm.weight"), f32(Lp + "output_query.LayerNorm.bias"), self.eps, stats=True)
            rec.update(hx=hx, z=z, f=f, s3=s3, m3=m3, r3=r3, k3=k3)
            S["layers"].append(rec)
        y = ops.gemm_nt(h, st.bf16_view(self.prefix + "linear.weight"), bias=f32(self.prefix + "linear.bias"))
        out, mo, ro = ops.layernorm(y, f32(self.prefix + "norm.weight"), f32(self.prefix + "norm.bias"), 1e-5, stats=True)
        if train:
            S.update(h_last=h, y=y, mo=mo, ro=ro)
            stash["qformer"] = S
        return out.view(B, Q, self.dl)

    # ---- backward --------------------------------------------------------------------------------------
    def backward_hip(self, dout: torch.Tensor, stash: dict, acc: bool):
        """dout [B*Q, llm_dim] bf16; deposits every parameter gradient into the flat grad buffer"""
        st, p, P, d, H, Fd, Q = self.store, self.prefix, self.prefix + "qformer.", self.d, self.H, self.F, self.Q
        S = stash.pop("qformer")
        B, Tk, km, enc2d = S["B"], S["Tk"], S["km"], S["enc2d"]
        f32, gv = st.master_view, st.grad_view
        scale = 64 ** -0.5

        def ln_bwd(x, mean, rstd, name, dy):
            return ops.layernorm_bwd(x, mean, rstd, f32(name + ".weight"), dy, dgamma=gv(name + ".weight"), dbeta=gv(name + ".bias"),
                                     accumulate=acc)

        def undrop(dy, key):
            """gradient through a hidden dropout: the same mask, recomputed (the residual branch keeps dy itself)"""
            return dy if key is None else ops.dropout(dy, *key)

        dy = ln_bwd(S["y"], S["mo"], S["ro"], p + "norm", dout)
        dh = self._lin_bwd(dy, S["h_last"], p + "linear.weight", p + "linear.bias", self.dl, d, acc, p + "linear.weight")
        d_enc = None   # dL/d(encoder states) [B*Tk, d_enc], only with a trainable encoder (need_dx)
        for l in reversed(range(self.L)):
            R = S["layers"][l]
            Lp = f"{P}encoder.layer.{l}."
            # feed-forward block
            ds3 = ln_bwd(R["s3"], R["m3"], R["r3"], Lp + "output_query.LayerNorm", dh)
            df = self._lin_bwd(undrop(ds3, R["k3"]), R["f"], Lp + "output_query.dense.weight", Lp + "output_query.dense.bias", d, Fd, acc,
                               Lp + "output_query.dense.weight")
            dz = ops.gelu_bwd(R["z"], df)
            dhx = self._lin_bwd(dz, R["hx"], Lp + "intermediate_query.dense.weight", Lp + "intermediate_query.dense.bias", Fd, d,
                                acc, Lp + "intermediate_query.dense.weight")
            dhx = self._add(dhx, ds3)  # residual into hx
            # cross-attention block
            if R["cross"] is not None:
                X = R["cross"]
                C = Lp + "crossattention."
                ds2 = ln_bwd(X["s2"], X["mc"], X["rc"], C + "output.LayerNorm", dhx)
                dc = self._lin_bwd(undrop(ds2, X["k2"]), X["c"], C + "output.dense.weight", C + "output.dense.bias", d, d, acc, C + "output.dense.weight")
                dqc = torch.empty_like(X["qc"])
                dkvc = torch.empty_like(X["kvc"])
                # (32 queries: the <= 64-query backward kernels read transposed copies of q / k / dO; ops.attn_bwd builds them)
                ops.attn_bwd(X["qc"], X["kvc"][:, :d], X["kvc"][:, d:], X["c"], dc, X["lse"], dqc,
                             dkvc[:, :d], dkvc[:, d:], B, Q, H, H, 64, False, scale, key_mask=km, Tk=Tk, drop=X["ad"])
                dh1 = self._lin_bwd(dqc, R["h1"], C + "attention.query.weight", C + "attention.query.bias", d, d, acc,
                                    C + "attention.query.weight")
                dx_enc = self._lin_bwd(dkvc, enc2d, C + "attention.key.weight", C + "attention.key.bias", 2 * d, self.d_enc, acc,
                                       (C + "attention.kv") if self.need_dx else None)
                if dx_enc is not None:   # every cross-attention layer reads the same encoder states: their gradients add up
                    d_enc = dx_enc if d_enc is None else self._add(d_enc, dx_enc)
                dh1 = self._add(dh1, ds2)
            else:
                dh1 = dhx
            # self-attention block
            A = Lp + "attention."
            ds1 = ln_bwd(R["s1"], R["m1"], R["r1"], A + "output.LayerNorm", dh1)
            da = self._lin_bwd(undrop(ds1, R["k1"]), R["a"], A + "output.dense.weight", A + "output.dense.bias", d, d, acc, A + "output.dense.weight")
            qkv = R["qkv"]
            dqkv = torch.empty_like(qkv)
            ops.attn_bwd(qkv[:, :d], qkv[:, d: 2 * d], qkv[:, 2 * d:], R["a"], da, R["lse"], dqkv[:, :d],
                         dqkv[:, d: 2 * d], dqkv[:, 2 * d:], B, Q, H, H, 64, False, scale, drop=R["ad"])
            dh = self._lin_bwd(dqkv, R["h"], A + "attention.query.weight", A + "attention.query.bias", 3 * d, d, acc,
                               A + "attention.qkv")
            dh = self._add(dh, ds1)
        # queries: h0 = LN(query) broadcast over the batch -> sum the batch, then LayerNorm backward
        dh = undrop(dh, S["k0"])
        dsum = torch.empty((Q * d,), dtype=torch.float32, device=dh.device)
        ops.colsum(dh.view(B, Q * d), dsum)
        dsum_bf = ops.cast_bf16(dsum).view(Q, d)
        dq0 = ln_bwd(S["q0"], S["m0"], S["r0"], P + "layernorm", dsum_bf)
        ops.cast_f32_(dq0, gv(p + "query"), accumulate=acc)
        return d_enc

    @staticmethod
    def _add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        """a += b on bf16 activations via the GEMM-free path: b is folded in as the `dres` of an identity... kept simple:
        a single fused kernel would be nicer; the tensors here are [B*Q, 768] (tiny)."""
        ops.add_(a, b)
        return a

    def forward(self, x, atts):
        return self.forward_hip(x, atts, None)
