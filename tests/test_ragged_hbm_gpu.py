"""GPU: the ragged buffer at the size the 288 GB part is for (`north_star`: "packs variable-length audio into ragged buffers sized for
288 GB HBM3E"; VERDICT r4 missing #4).

`batcher.frames_for_hbm()` (~61 k real tokens) as `max_frame_length` with `dataset_config.frame_budget=sum` forms ONE batch of ~250
clips where the reference's 12 000-frame window forms ~6.  This test runs that batch through the ragged path (`varlen_encoder` +
`varlen`: 200 k packed encoder rows, ~60 k packed LLM tokens: every buffer 5x the headline batch's -- GEMM operands with M x ld in
the GB range, attention over a 60 k-token packed axis, int32 segment tables, the labelled-row gather) at true Whisper-large-v3 /
Llama-3-8B widths and REDUCED depth (2 + 2 layers: the stash of 32 layers is what `frames_for_hbm` sizes; tools/ragged_bench.py
`ragged_sum_hbm` runs it at full depth), and checks the size-independent property the domain offers: the loss is a token mean and
the gradients are token sums, so

    loss(big batch)  = sum_g n_g loss(group g) / sum_g n_g          (n_g = labelled tokens of group g)
    grad(big batch)  = sum_g n_g grad(group g) / sum_g n_g

over the 12 000-budget groups of the SAME clips (reference knob: datasets/speech_dataset_large.py:244-263)."""
import pytest
import torch

from tests import golden_util as G

pytestmark = pytest.mark.gpu


def _clips(n_clips, seed):
    from slam_llm_amd import batcher
    g = torch.Generator().manual_seed(seed)
    secs = torch.rand(n_clips, generator=g) * 28 + 2
    samples = []
    for s_ in secs.tolist():
        n = int(s_ * 16000) // 160 * 160
        alen = batcher.whisper_audio_length(n, 5, pad_to_30s=False)
        A = int(torch.randint(8, 129, (1,), generator=g))
        audio = (torch.randn(n, generator=g) * 0.1).clamp_(-1, 1)
        samples.append(batcher.make_sample(audio, torch.randint(3, 128000, (16,), generator=g).tolist(),
                                           torch.randint(3, 128000, (A - 1,), generator=g).tolist(), 2, alen))
    return samples


@pytest.mark.timeout(1800)
def test_hbm_sized_ragged_batch_equals_its_12k_budget_groups(dev):
    from slam_llm_amd import batcher
    from slam_llm_amd.model import SlamHipModel, make_config
    budget = batcher.frames_for_hbm()
    assert 55_000 < budget < 70_000, budget
    samples = _clips(420, seed=4321)
    big = next(iter(batcher.dynamic_batches(iter(samples), budget, budget="sum")))
    n_tok = sum(len(s["input_ids"]) for s in big)
    assert budget - 500 < n_tok <= budget and len(big) > 180, (n_tok, len(big))
    groups = list(batcher.dynamic_batches(iter(big), 12000, budget="sum"))
    assert len(groups) >= 5 and sum(len(g_) for g_ in groups) == len(big)
    cfg = make_config("whisper-large-v3", "llama-3-8b", enc_layers=2, llm_layers=2, lora_r=16, lora_alpha=32,
                      lora_targets=("q_proj", "v_proj"), lora_dropout=0.0, pad_or_trim=False, varlen=True, varlen_encoder=True)
    model = SlamHipModel(cfg, dev).init_random(42)
    model.train()

    def run(grp):
        b = batcher.collate(grp, 0, left_pad_prompt=False, pad_or_trim=False)
        n_lab = int((b["labels"][:, 1:] != -100).sum())
        gb = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
        model.store.grad.zero_()
        out, acc = model(**gb)
        out.loss.backward()
        torch.cuda.synchronize()
        return float(out.loss.detach()), float(acc), n_lab, model.store.grad.clone()

    torch.cuda.reset_peak_memory_stats()
    loss_big, acc_big, n_big, g_big = run(big)
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    assert torch.isfinite(g_big).all() and loss_big == loss_big
    parts = [run(g_) for g_ in groups]
    n_sum = sum(p[2] for p in parts)
    assert n_sum == n_big
    loss_w = sum(p[0] * p[2] for p in parts) / n_sum
    acc_w = sum(p[1] * p[2] for p in parts) / n_sum
    g_w = sum(p[3].double() * p[2] for p in parts) / n_sum
    # fp32 token sums in a different order + bf16 products tiled differently (the M = 60 k products take other kernels than the
    # M = 12 k ones): loss to 2e-3 (measured: see profiles/r05_ragged.md), accuracy to one token, gradients cosine >= 0.9995 per tensor
    assert abs(loss_big - loss_w) <= 2e-3, (loss_big, loss_w)
    assert abs(acc_big - acc_w) <= 1.5 / n_big, (acc_big, acc_w)
    worst = 1.0
    for name, (off, n, _) in model.store.offsets.items():
        a, b_ = g_big[off:off + n].double(), g_w[off:off + n]
        cs = float((a * b_).sum() / (a.norm() * b_.norm() + 1e-300))
        worst = min(worst, cs)
        G.floor_check(cs, 0.9995, f"{name}: cosine {cs} between the {budget}-token batch and its 12 000-token groups")
        assert abs(float(a.norm()) - float(b_.norm())) <= 1e-2 * float(b_.norm()), name
    print(f"hbm-sized ragged batch: {len(big)} clips, {n_tok} tokens (budget {budget}), {len(groups)} groups at 12 000; loss {loss_big:.5f} vs "
          f"{loss_w:.5f}, worst gradient cosine {worst:.6f}, peak HBM {peak:.1f} GB at 2 + 2 layers")
