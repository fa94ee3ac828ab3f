"""GPU: the training step captured in a HIP graph (slam_llm_amd.train.GraphedTrainStep) and the three device-side
pieces that took the host out of it: slam_label_rows (labelled-row selection against a static bound), slam_adamw_step_dev (lr and bias
corrections from device memory) and the dropout salt (fresh masks on every replay).  Reference loop body: utils/train_utils.py:112-169."""
import numpy as np
import pytest
import torch

from oracle import slam_oracle as O

pytestmark = pytest.mark.gpu


def _model_and_batches(dev, lora_dropout=0.0, seed=42, n_batches=2):
    from slam_llm_amd.model import SlamHipModel
    cfg = O.make_config()
    W = O.init_weights(cfg, seed=seed)
    model = SlamHipModel(dict(cfg, lora_dropout=lora_dropout), dev).load_weights(W)
    model.train()
    batches = []
    for i in range(n_batches):
        audio = O.synth_audio(2, 2.0, seed=1234 + i)
        ob = O.synth_batch(cfg, audio, prompt_len=6, answer_lens=(5, 9), seed=1236 + i, left_pad=True, pad_to_30s=False)
        batches.append({k: v.to(dev) for k, v in ob.items()})
    return model, batches


@pytest.mark.parametrize("M,frac,cap_extra", [(380, 0.17, 0), (11780, 0.17, 64), (70000, 0.5, 1000), (1000, 0.0, 5), (1024, 1.0, 0), (3000, 0.3, -100)])
def test_label_rows_matches_torch(dev, M, frac, cap_extra):
    """rows / tsel / inv / count of slam_label_rows against the torch formulation it replaces (argsort of `targets < 0`, scatter of
    arange): bit-exact integer work; positions past the count are -1; a bound BELOW the count truncates and still reports the count."""
    from slam_llm_amd import ops
    g = torch.Generator().manual_seed(M)
    t = torch.randint(0, 1000, (M,), generator=g, dtype=torch.int32)
    t[torch.rand(M, generator=g) >= frac] = -1
    n = int((t >= 0).sum())
    cap = max(1, min(M, n + cap_extra))
    rows, tsel, inv, cnt = ops.label_rows(t.to(dev), cap)
    assert int(cnt) == n
    want_rows = torch.nonzero(t >= 0).flatten().to(torch.int32)
    k = min(n, cap)
    assert torch.equal(rows.cpu()[:k], want_rows[:k]) and torch.equal(tsel.cpu()[:k], t[want_rows[:k].long()])
    assert bool((rows.cpu()[k:] == -1).all()) and bool((tsel.cpu()[k:] == -1).all())
    want_inv = torch.full((M,), -1, dtype=torch.int32)
    want_inv[want_rows[:k].long()] = torch.arange(k, dtype=torch.int32)
    assert torch.equal(inv.cpu(), want_inv)


def test_adamw_step_dev_equals_adamw_step(dev):
    """the fused AdamW with lr / bias corrections read from device memory is the same kernel: bit-identical parameters, moments and
    bf16 copy after three steps with a changing lr"""
    from slam_llm_amd import ops
    n = 100_003
    g = torch.Generator(device=dev).manual_seed(3)
    p0 = torch.randn(n, generator=g, device=dev)
    runs = []
    for mode in ("host", "dev"):
        p, m, v, pb = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.bfloat16, device=dev)
        gg = torch.Generator(device=dev).manual_seed(4)
        for step, lr in enumerate((0.0, 1e-3, 5e-4), start=1):
            grad = torch.randn(n, generator=gg, device=dev)
            if mode == "host":
                ops.adamw_step(p, grad, m, v, pb, lr, 0.9, 0.999, 1e-8, 0.01, step)
            else:
                hh = torch.zeros(3, dtype=torch.float32)
                ops.adamw_hyper(lr, 0.9, 0.999, step, hh)
                ops.adamw_step_dev(p, grad, m, v, pb, hh.to(dev), 0.9, 0.999, 1e-8, 0.01)
        runs.append((p, m, v, pb))
    for a, b in zip(*runs):
        assert torch.equal(a, b)


def test_static_label_bound_equals_exact_count(dev):
    """a static bound above the real count adds zero rows with ignored targets behind the labelled ones: loss, accuracy and every
    gradient equal the exact-count path (the default: count read back from the device) to fp32 summation order"""
    model, batches = _model_and_batches(dev)
    res = []
    for cap in (None, 64, batches[0]["input_ids"].numel()):
        model.llm.label_rows_cap = cap
        for p in model.store.params.values():
            p.grad = None
        out, acc = model(**{k: v.clone() for k, v in batches[0].items()})
        out.loss.backward()
        res.append((float(out.loss), float(acc), model.store.grad.clone()))
        if cap:
            assert int(model.llm.last_label_count) == int((batches[0]["labels"][:, 1:] >= 0).sum())
    for loss, acc, grad in res[1:]:
        assert abs(loss - res[0][0]) <= 2e-6 * abs(res[0][0]) and acc == res[0][1]
        scale = float(res[0][2].abs().max())
        assert float((grad - res[0][2]).abs().max()) <= 2e-5 * scale


def test_graphed_step_is_bit_identical_to_eager(dev):
    """5 steps over two alternating batches, LambdaLR warm-up (lr changes every step), lora_dropout 0: the captured step (2 eager
    warm-up calls, capture on the 3rd, replays afterwards) must reproduce the eager loop bit for bit -- losses, accuracies, trainable
    parameters, Adam moments -- with the same static label bound on both sides."""
    from slam_llm_amd.model import SlamAdamW
    from slam_llm_amd.train import GraphedTrainStep, lr_lambda, train_step
    runs = []
    for graphed in (False, True):
        model, batches = _model_and_batches(dev)
        model.llm.label_rows_cap = 64
        opt = SlamAdamW(model, lr=1e-2, weight_decay=0.01)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: lr_lambda(s, 3, 10))
        stepper = GraphedTrainStep(model, opt, sched, label_rows_cap=64, warmup=2) if graphed else None
        losses = []
        for i in range(6):
            b = {k: v.clone() for k, v in batches[i % 2].items()}
            loss, acc = stepper(b) if graphed else train_step(model, b, opt, sched)
            losses.append((float(loss), float(acc)))
        torch.cuda.synchronize()
        runs.append((losses, model.store.flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt._step))
        if graphed:
            assert stepper.replays == 4 and stepper.eager_steps == 2
            # a batch of another shape falls back to the eager step and the captured one keeps working afterwards
            other = {k: (torch.cat([v, v]) if v.dim() else v) for k, v in batches[0].items()}
            stepper(other)
            assert stepper.eager_steps == 3
            stepper({k: v.clone() for k, v in batches[1].items()})
            assert stepper.replays == 5
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    for a, b in zip(runs[0][1:4], runs[1][1:4]):
        assert torch.equal(a, b)
    assert runs[0][4] == runs[1][4] == 6


def test_graphed_step_draws_fresh_dropout_masks(dev):
    """lora_dropout 0.3, lr 0 (the parameters never move): replays of the SAME batch must see different masks -- the captured kernel
    arguments are frozen, the salt word the kernels XOR into their seed is not -- and the masks of forward and backward of one replay
    must agree (the gradient of a replay equals the eager gradient under the same salt)."""
    from slam_llm_amd import ops
    from slam_llm_amd.model import SlamAdamW
    from slam_llm_amd.train import GraphedTrainStep
    model, batches = _model_and_batches(dev, lora_dropout=0.3)
    opt = SlamAdamW(model, lr=0.0)
    stepper = GraphedTrainStep(model, opt, None, label_rows_cap=64, warmup=1)
    losses = [float(stepper({k: v.clone() for k, v in batches[0].items()})[0]) for _ in range(5)]
    assert stepper.replays == 4
    assert len(set(losses[1:])) == 4, losses        # four replays, four different masks
    assert max(losses) - min(losses) < 0.5          # ... of the same model on the same batch
    assert int(stepper.salt) != 0


def test_graphed_step_reports_a_label_bound_that_was_too_small(dev):
    from slam_llm_amd.model import SlamAdamW
    from slam_llm_amd.train import GraphedTrainStep
    model, batches = _model_and_batches(dev)
    opt = SlamAdamW(model, lr=1e-3)
    n = int((batches[0]["labels"][:, 1:] >= 0).sum())
    stepper = GraphedTrainStep(model, opt, None, label_rows_cap=max(1, n - 3), warmup=1)
    stepper({k: v.clone() for k, v in batches[0].items()})
    stepper({k: v.clone() for k, v in batches[0].items()})       # captured + replayed with a bound below the count
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="labelled rows"):
        stepper({k: v.clone() for k, v in batches[0].items()})


@pytest.mark.parametrize("graphed", [False, True])
def test_lora_side_stream_is_bit_identical(dev, graphed):
    """VERDICT r5 next #5a: the adapter-gradient products (dA / dB grams + reduces) on a side stream, joined one layer late, are the
    same kernels on the same operands with the same fixed-order reductions: losses, gradients and parameters after 4 steps equal the
    inline order bit for bit -- eager and inside the captured step (the side stream becomes a parallel branch of the graph)."""
    from slam_llm_amd import model as M
    from slam_llm_amd.model import SlamAdamW
    from slam_llm_amd.train import GraphedTrainStep, train_step
    runs = []
    try:
        for side in (False, True):
            M.LORA_SIDE_STREAM = side
            model, batches = _model_and_batches(dev, lora_dropout=0.1)
            model.llm.label_rows_cap = 64
            opt = SlamAdamW(model, lr=1e-2)
            stepper = GraphedTrainStep(model, opt, None, label_rows_cap=64, warmup=1) if graphed else None
            losses, g0 = [], None
            for i in range(4):
                b = {k: v.clone() for k, v in batches[i % 2].items()}
                if graphed:
                    loss, _ = stepper(b)
                else:
                    out, _ = model(**b)
                    out.loss.backward()
                    if i == 0:
                        torch.cuda.synchronize()
                        g0 = model.store.grad.clone()
                    opt.step()
                    opt.zero_grad()
                    loss = out.loss.detach()
                losses.append(float(loss))
            torch.cuda.synchronize()
            runs.append((losses, g0, model.store.flat.clone()))
    finally:
        M.LORA_SIDE_STREAM = False
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    if not graphed:
        assert torch.equal(runs[0][1], runs[1][1])
    assert torch.equal(runs[0][2], runs[1][2])


@pytest.mark.timeout(600)
def test_graphed_step_is_bit_identical_at_true_widths(dev):
    """the capture at the shapes the bench runs (Whisper-large-v3 x 1 -> Llama-3-8B x 1 at true widths, 8 x 30 s clips, raw audio in, T = 380:
    GPU log-mel, the persistent and the 4-wave GEMM kernels with their registered workspace, chunk-sized lm_head over the label rows, the
    pruned last layer): 4 steps eager vs 1 eager + 3 replays, lora_dropout 0 -- bit-identical losses and parameters."""
    from slam_llm_amd.model import SlamAdamW, SlamHipModel, make_config
    from slam_llm_amd.train import GraphedTrainStep, train_step
    cfg = make_config("whisper-large-v3", "llama-3-8b", enc_layers=1, llm_layers=1, lora_r=16, lora_alpha=32,
                      lora_targets=("q_proj", "v_proj"), lora_dropout=0.0)
    W = O.init_weights(cfg, seed=42)
    B = 8
    audio = O.synth_audio(B, 30.0, seed=1234)
    ob = O.synth_batch(cfg, audio, prompt_len=16, answer_lens=(64,), seed=1236, left_pad=False, pad_to_30s=True)
    gb = {k: v.to(dev) for k, v in ob.items() if k != "audio_mel"}
    gb["audio"] = audio.to(dev)
    runs = []
    for graphed in (False, True):
        model = SlamHipModel(dict(cfg), dev).load_weights(W)
        model.train()
        model.llm.label_rows_cap = B * 64
        opt = SlamAdamW(model, lr=1e-3)
        stepper = GraphedTrainStep(model, opt, None, label_rows_cap=B * 64, warmup=1) if graphed else None
        losses = []
        for _ in range(4):
            b = {k: v.clone() for k, v in gb.items()}
            loss, _ = stepper(b) if graphed else train_step(model, b, opt)
            losses.append(float(loss))
        torch.cuda.synchronize()
        if graphed:
            assert stepper.replays == 3 and stepper.eager_steps == 1
        runs.append((losses, model.store.flat.clone()))
        del model, opt, stepper
        torch.cuda.empty_cache()
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1])
