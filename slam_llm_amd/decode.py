"""Search bookkeeping for `SlamHipModel.generate` -- the host side of HF `GenerationMixin.generate` as the reference
calls it (src/slam_llm/models/slam_model.py:438-452: num_beams=4, do_sample=False, max_new_tokens=200, min_length=1,
length_penalty=1.0, early_stopping unset, one eos id, num_return_sequences=1).

Everything here is small integer/float tensor work on the device the logits live on (a handful of [batch, 2*beams]
tensors per step); the model work -- prefill, KV-cache decode steps -- is behind `step_fn` on the HIP path.  One
host sync per generated token (the stop test), like HF.

step_fn(tokens [R, t] int64, src_rows [R] int64 | None) -> next-token logits [R, V] fp32
    R = batch (greedy) or batch*num_beams; row r of this call continues row src_rows[r] of the previous call
    (None = identity).  t == 0 on the first call (the prompt is embeddings only, the token history starts empty).
"""
from __future__ import annotations

import torch

GENERATE_KEYS = ("max_new_tokens", "num_beams", "do_sample", "min_length", "top_p", "repetition_penalty",
                 "length_penalty", "temperature", "eos_token_id", "pad_token_id", "bos_token_id", "max_length")
NEG = -1.0e9


def greedy_search(step_fn, batch_size: int, max_new_tokens: int, eos: int, pad: int, min_length: int, device):
    """argmax decode: eos is masked while fewer than `min_length` tokens exist; a finished row emits `pad`; stops
    when every row has emitted eos or max_new_tokens is reached.  Returns [batch, steps] int64."""
    toks = torch.zeros((batch_size, 0), dtype=torch.int64, device=device)
    alive = torch.ones(batch_size, dtype=torch.bool, device=device)
    while True:
        logits = step_fn(toks, None)
        if toks.shape[1] < min_length:
            logits = logits.clone()
            logits[:, eos] = -float("inf")
        nxt = torch.where(alive, logits.argmax(-1), torch.full((batch_size,), pad, dtype=torch.int64, device=device))
        toks = torch.cat([toks, nxt[:, None]], dim=1)
        alive = alive & (nxt != eos)
        if toks.shape[1] >= max_new_tokens or not bool(alive.any()):
            return toks


def _take(t: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """t [B, n, ...] gathered along dim 1 by idx [B, m]"""
    while idx.dim() < t.dim():
        idx = idx.unsqueeze(-1)
    return torch.take_along_dim(t, idx, dim=1)


def beam_search(step_fn, batch_size: int, num_beams: int, max_new_tokens: int, eos: int, pad: int, min_length: int,
                length_penalty: float, device):
    """Beam search with HF's semantics: per item `num_beams` running and `num_beams` finished hypotheses; each step
    ranks the best 2*num_beams continuations of the running set; the non-terminated ones refill the running set; a
    terminated one (eos, or the max_new_tokens-th token) is admitted to the finished set -- scored
    sum_logprob / length**length_penalty -- only from the first num_beams ranks.  An item closes when
    best_running_sum / cur_len**length_penalty can no longer beat its worst finished score.  Returns the best
    finished hypothesis per item, [batch, longest], filled with `pad or eos` (HF's `pad_token_id or eos` quirk)."""
    B, nb, L, K = batch_size, num_beams, max_new_tokens, 2 * num_beams
    fill = pad or eos
    i64 = dict(dtype=torch.int64, device=device)
    run_seq = torch.full((B, nb, L), fill, **i64)
    run_score = torch.zeros((B, nb), dtype=torch.float32, device=device)
    run_score[:, 1:] = NEG
    fin_seq = run_seq.clone()
    fin_score = torch.full((B, nb), NEG, dtype=torch.float32, device=device)
    fin_flag = torch.zeros((B, nb), dtype=torch.bool, device=device)
    fin_len = torch.zeros((B, nb), **i64)
    item_open = torch.ones((B, 1), dtype=torch.bool, device=device)
    first_ranks = (torch.arange(K, device=device) < nb)[None, :]
    row_base = torch.arange(B, **i64)[:, None] * nb
    src_rows = torch.arange(B * nb, **i64)
    t = 0
    while True:
        logits = step_fn(run_seq[:, :, :t].reshape(B * nb, t), src_rows)
        lp = torch.log_softmax(logits.float(), dim=-1)
        V = lp.shape[-1]
        if t < min_length:
            lp[:, eos] = -float("inf")
        acc = (lp.view(B, nb, V) + run_score[:, :, None]).view(B, nb * V)
        top_lp, top_idx = torch.topk(acc, K)
        src, tok = top_idx // V, top_idx % V
        cand = _take(run_seq, src)
        cand[:, :, t] = tok
        hits = (tok == eos) if t + 1 < L else torch.ones_like(tok, dtype=torch.bool)
        # running set
        run_lp = top_lp + hits.float() * NEG
        sel = torch.topk(run_lp, nb)[1]
        run_seq, run_score = _take(cand, sel), _take(run_lp, sel)
        src_rows = (_take(src, sel) + row_base).view(-1)
        # finished set
        just = hits & first_ranks
        sc = top_lp / ((t + 1) ** length_penalty)
        sc = sc + (~item_open).float() * NEG
        sc = sc + (~just).float() * NEG
        m_score = torch.cat([fin_score, sc], dim=1)
        idx = torch.topk(m_score, nb)[1]
        fin_seq = _take(torch.cat([fin_seq, cand], dim=1), idx)
        fin_score = _take(m_score, idx)
        fin_flag = _take(torch.cat([fin_flag, just], dim=1), idx)
        fin_len = _take(torch.cat([fin_len, torch.full((B, K), t + 1, **i64)], dim=1), idx)
        t += 1
        best = run_score[:, :1] / (t ** length_penalty)
        worst = torch.where(fin_flag, fin_score.min(dim=1, keepdim=True)[0], torch.full_like(fin_score, NEG))
        item_open = item_open & (best > worst).any(dim=-1, keepdim=True)
        if t >= L or not bool(item_open.any()):   # the single host sync of the step
            break
    out_len = int(fin_len[:, 0].max())
    return fin_seq[:, 0, :out_len].contiguous()
