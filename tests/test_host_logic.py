"""CPU: host-side logic of the product (batcher, collators, LR schedule, FLOP model) against the oracle / fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import slam_oracle as O
from slam_llm_amd import batcher
from slam_llm_amd.train import lr_lambda
from tests import golden_util as G


def test_dynamic_batcher_membership_matches_reference_fixture():
    fx = G.load("batcher")  # produced by the reference's own window_class / MultiTaskDynamicBatchDataset
    ci = 0
    while f"lens.{ci}" in fx.files:
        lens = [int(x) for x in fx[f"lens.{ci}"]]
        samples = [{"input_ids": torch.zeros(n, dtype=torch.int64), "idx": i} for i, n in enumerate(lens)]
        groups = list(batcher.dynamic_batches(samples, int(fx[f"mfl.{ci}"])))
        assert [len(g) for g in groups] == [int(x) for x in fx[f"group_sizes.{ci}"]]
        assert [s["idx"] for g in groups for s in g] == list(range(len(lens)))
        ci += 1
    assert ci == 4


def _samples():
    g = torch.Generator().manual_seed(0)
    out_b, out_o = [], []
    for alen, pl, al in [(20, 6, 5), (20, 6, 9), (12, 4, 1)]:
        pids = torch.randint(3, 100, (pl,), generator=g).tolist()
        aids = torch.randint(3, 100, (al,), generator=g).tolist()
        audio = torch.randn(1600 * (alen // 4), generator=g)
        out_b.append(batcher.make_sample(audio, pids, aids, 2, alen))
        out_o.append(O.make_sample(alen, pids, aids, 2))
    return out_b, out_o


def test_collators_match_oracle_restatement_of_reference():
    sb, so = _samples()
    for left in (True, False):
        got = batcher.collate(sb, pad_token_id=2, left_pad_prompt=left)
        ref = (O.collate_left_pad if left else O.collate_right_pad)(so, pad_id=2)
        for k in ("input_ids", "labels", "attention_mask", "modality_mask"):
            assert torch.equal(got[k], ref[k]), (left, k)
        assert got["audio"].shape[0] == 3 and got["audio_len"].tolist() == [len(s["audio"]) for s in sb]


def test_mel_collator_emits_the_reference_post_mask():
    """`audio_mel_post_mask` as the reference collators build it (speech_dataset.py:246-249,
    speech_dataset_large.py:198-200) -- ragged clips (pad_or_trim off): ones over each clip's own (frames + 1) // 2 encoder frames
    out of (Tmax + 1) // 2; pad_or_trim on: every clip is 3000 mel frames."""
    lens = [16000 * 3 + 77, 16000 * 5, 16000 * 2 + 161]
    samples = [batcher.make_sample(torch.zeros(n), [5, 6], [7, 8], 2, batcher.whisper_audio_length(n, 5, pad_to_30s=False)) for n in lens]
    got = batcher.collate(samples, pad_token_id=2, left_pad_prompt=True, pad_or_trim=False)
    # the oracle's collator is the reference's: feed it mels of the lengths whisper.log_mel_spectrogram gives (n // 160 frames)
    so = [O.make_sample(s["audio_length"], [5, 6], [7, 8], 2) for s in samples]
    ref = O.collate_left_pad(so, pad_id=2, mels=[torch.zeros(n // 160, 80) for n in lens])
    assert torch.equal(got["audio_mel_post_mask"], ref["audio_mel_post_mask"])
    assert got["audio_mel_post_mask"].shape == (3, (500 + 1) // 2) and got["audio_mel_post_mask"].sum(1).tolist() == [150.0, 250.0, 101.0]
    got = batcher.collate(samples, pad_token_id=2, left_pad_prompt=False)          # pad_or_trim (default): 3000 frames each
    assert got["audio_mel_post_mask"].shape == (3, 1500) and bool(got["audio_mel_post_mask"].all())
    raw = batcher.collate(samples, pad_token_id=2, left_pad_prompt=True, input_type="raw")
    assert "audio_mel_post_mask" not in raw and raw["audio_mask"].shape == (3, max(lens))


def test_golden_batches_are_reproduced_by_product_collator():
    """the batch stored in the step fixtures (built through the oracle, equal to the reference layout) is
    reproduced by the product's make_sample + collate"""
    from oracle.make_golden_cases import CASES
    for name, case in CASES.items():
        fx = G.load(name)
        cfg = case["cfg"]
        g = torch.Generator().manual_seed(1236)
        audio = torch.from_numpy(fx["audio"])
        samples = []
        for i in range(audio.shape[0]):
            al = case["answer_lens"][i % len(case["answer_lens"])]
            pids = torch.randint(3, cfg["vocab"], (6,), generator=g).tolist()
            aids = torch.randint(3, cfg["vocab"], (al - 1,), generator=g).tolist()
            alen = batcher.whisper_audio_length(audio.shape[1], cfg["ds_rate"], pad_to_30s=False)
            samples.append(batcher.make_sample(audio[i], pids, aids, 2, alen))
        got = batcher.collate(samples, pad_token_id=2, left_pad_prompt=case["left_pad"])
        for k in ("input_ids", "labels", "attention_mask", "modality_mask"):
            assert np.array_equal(got[k].numpy(), fx["batch." + k]), (name, k)


def test_whisper_audio_length():
    assert batcher.whisper_audio_length(16000 * 10) == 300  # padded to 30 s by the recipe (speech_dataset.py:101-105)
    assert batcher.whisper_audio_length(32000, pad_to_30s=False) == 20


def test_lr_schedule_matches_reference_lambda():
    for s in (0, 1, 999, 1000, 1001, 50000, 100000, 100001):
        assert lr_lambda(s, 1000, 100000) == O.lr_lambda(s, 1000, 100000)
    assert lr_lambda(0, 1000, 100000) == 0  # first optimizer step is a no-op on the parameters


def test_flop_model_matches_survey_worked_example():
    import bench
    from slam_llm_amd.model import make_config
    cfg = make_config("whisper-large-v3", "llama-3-8b", lora_r=16, lora_alpha=32)
    fl = bench.algorithmic_flops_per_clip(cfg, T=380, Ta=300, n_frames=3000)
    # enc 2.27e12, proj 3.9e10, llm 1.15e13, lora 1.6e10, mel 1.1e9 -> 1.38e13 per 30 s clip
    assert abs(fl["enc"] / 2.27e12 - 1) < 0.02 and abs(fl["proj"] / 3.9e10 - 1) < 0.03
    assert abs(fl["llm"] / 1.15e13 - 1) < 0.02 and abs(fl["total"] / 1.38e13 - 1) < 0.02
    assert batcher.frames_for_hbm() > 60000


@pytest.mark.parametrize("scale", [24.0, 5.0])
def test_decode_bookkeeping_matches_reference_generate(scale):
    """the product's beam / greedy bookkeeping (slam_llm_amd/decode.py), driven on CPU by the oracle's fp32 model
    step, reproduces the token ids the reference's generate() produced (tests/golden/generate.npz)."""
    import torch.nn.functional as F
    from oracle import slam_oracle as O
    from oracle.make_golden_cases import GENERATE_CASE as C
    from slam_llm_amd import decode
    from tests.test_oracle_golden import GEN_RUNS, gen_key, generate_case_weights
    fx = G.load("generate")
    cfg, W = C["cfg"], generate_case_weights(scale)
    batch = {k[len("batch."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("batch.")}
    enc = O.whisper_encoder(W, cfg, batch["audio_mel"].permute(0, 2, 1))
    proj = O.projector_concat(W, enc, cfg["ds_rate"])
    emb_w = W["llm.base_model.model.model.embed_tokens.weight"]
    embeds = O.embed_splice(emb_w, batch["input_ids"].clone(), batch["modality_mask"].bool(), proj)
    mask = batch["attention_mask"].long()
    B = embeds.shape[0]

    def step_fn(tokens, src_rows):
        item = torch.arange(tokens.shape[0]) // (tokens.shape[0] // B)
        x = torch.cat([embeds[item], F.embedding(tokens, emb_w)], dim=1)
        m = torch.cat([mask[item], torch.ones_like(tokens)], dim=1)
        return O.llama_forward(W, cfg, x, m, None, position_ids=O.generate_position_ids(m))[1][:, -1, :]

    eos = int(fx[f"s{scale}.eos"])
    for nb, lp, pad, rp in GEN_RUNS:
        if nb == 1:
            got = decode.greedy_search(step_fn, B, C["max_new_tokens"], eos, pad, 1, "cpu", rp)
        else:
            got = decode.beam_search(step_fn, B, nb, C["max_new_tokens"], eos, pad, 1, lp, "cpu", rp)
        want = fx[gen_key(scale, nb, lp, pad, rp)]
        assert tuple(got.shape) == want.shape and (got.numpy() == want).all(), (nb, lp, pad, got, want)


def test_sampling_decode_bookkeeping_matches_reference_generate():
    """do_sample=True: the product's warpers + multinomial bookkeeping (greedy-sample and beam-sample), driven on CPU by the
    oracle's fp32 model step and seeded like the fixture run, reproduces the tokens the reference's generate(do_sample=True)
    drew (tests/golden/generate_sample.npz, written by oracle/make_golden_sample.py from the reference + HF)."""
    import torch.nn.functional as F
    from oracle import slam_oracle as O
    from oracle.make_golden_cases import GENERATE_CASE as C
    from oracle.make_golden_sample import RUNS, key
    from slam_llm_amd import decode
    from tests.test_oracle_golden import generate_case_weights
    fx, fg = G.load("generate_sample"), G.load("generate")
    cfg, W = C["cfg"], generate_case_weights(float(fx["scale"]))
    batch = {k[len("batch."):]: torch.from_numpy(fg[k]) for k in fg.files if k.startswith("batch.")}
    enc = O.whisper_encoder(W, cfg, batch["audio_mel"].permute(0, 2, 1))
    proj = O.projector_concat(W, enc, cfg["ds_rate"])
    emb_w = W["llm.base_model.model.model.embed_tokens.weight"]
    embeds = O.embed_splice(emb_w, batch["input_ids"].clone(), batch["modality_mask"].bool(), proj)
    mask = batch["attention_mask"].long()
    B = embeds.shape[0]

    def step_fn(tokens, src_rows):
        item = torch.arange(tokens.shape[0]) // (tokens.shape[0] // B)
        x = torch.cat([embeds[item], F.embedding(tokens, emb_w)], dim=1)
        m = torch.cat([mask[item], torch.ones_like(tokens)], dim=1)
        return O.llama_forward(W, cfg, x, m, None, position_ids=O.generate_position_ids(m))[1][:, -1, :]

    eos = int(fx["eos"])
    for nb, temp, tk, tp, rp, seed in RUNS:
        sample = dict(temperature=temp, top_k=tk, top_p=tp)
        torch.manual_seed(seed)
        if nb == 1:
            got = decode.greedy_search(step_fn, B, C["max_new_tokens"], eos, 1, 1, "cpu", rp, sample)
        else:
            got = decode.beam_search(step_fn, B, nb, C["max_new_tokens"], eos, 1, 1, 1.0, "cpu", rp, sample)
        want = fx[key(nb, temp, tk, tp, rp, seed)]
        assert tuple(got.shape) == want.shape and (got.numpy() == want).all(), (nb, temp, tk, tp, rp, got, want)
        # and the oracle restatement, seeded the same way
        torch.manual_seed(seed)
        mine = O.slam_generate(W, cfg, {k: v.clone() for k, v in batch.items()}, max_new_tokens=C["max_new_tokens"], num_beams=nb,
                               eos=eos, pad=1, repetition_penalty=rp, sample=sample)
        assert (mine.numpy() == want).all()


def test_sampling_warpers():
    from slam_llm_amd import decode
    s = torch.tensor([[2.0, 1.0, 0.0, -1.0, -3.0]])
    assert torch.equal(decode.warp_scores(s, 2.0), s / 2.0)
    assert torch.isinf(decode.warp_scores(s, top_k=2)[0, 2:]).all() and torch.equal(decode.warp_scores(s, top_k=2)[0, :2], s[0, :2])
    p = torch.softmax(s, -1)[0]
    kept = ~torch.isinf(decode.warp_scores(s, top_p=0.9)[0])          # smallest head set with cumulative prob > 0.9
    assert kept.tolist() == [True, True, True, False, False] and float(p[:2].sum()) < 0.9 <= float(p[:3].sum())
    assert (~torch.isinf(decode.warp_scores(s, top_p=1e-6)[0])).tolist() == [True, False, False, False, False]


def test_inference_collator_reproduces_reference_layout():
    """inference-mode samples ([audio, prompt], no labels) through the product collator == the batch of
    tests/golden/generate.npz (built by the oracle's restatement of speech_dataset.py:120-134 + :216-273)."""
    fx = G.load("generate")
    ids, am, mm = (torch.from_numpy(fx["batch." + k]) for k in ("input_ids", "attention_mask", "modality_mask"))
    samples = []
    for b in range(ids.shape[0]):
        row = ids[b][am[b].bool()]
        alen = int(mm[b].sum())
        s = batcher.make_sample(torch.zeros(1600), row[alen:].tolist(), None, 2, alen)
        s.update(key=f"utt{b}", target=f"ref {b}")
        samples.append(s)
    out = batcher.collate(samples, pad_token_id=2, left_pad_prompt=True)
    assert torch.equal(out["attention_mask"], am.bool()) and torch.equal(out["modality_mask"], mm.bool())
    assert torch.equal(out["input_ids"].masked_fill(mm.bool(), 0), ids.masked_fill(mm.bool(), 0))
    assert "labels" not in out and out["keys"] == ["utt0", "utt1", "utt2"] and out["targets"][2] == "ref 2"


def _riff(pcm: np.ndarray, rate=16000) -> bytes:
    import struct
    data = pcm.astype("<i2").tobytes()
    return (b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16)
            + b"data" + struct.pack("<I", len(data)) + data)


def test_kaldi_ark_multitask_dataset_matches_reference_layout(tmp_path):
    """aispeech_asr input format: wav.ark entries addressed as path:offset, multitask.jsonl +
    multiprompt.jsonl; samples / right-padding collator == the oracle's restatement of speech_dataset_large.py:62-233;
    rank x worker sharding and the max_audio_length filter (:80-93)."""
    import json
    from types import SimpleNamespace
    from slam_llm_amd.dataset import MultiTaskDatasetRaw, get_speech_dataset, load_ark_wav
    g = torch.Generator().manual_seed(0)
    clips = [(torch.randn(n, generator=g) * 3000).round().clamp(-32768, 32767).numpy().astype(np.int16) for n in (16000, 4000, 24000, 16000 * 31)]
    ark, lines = tmp_path / "wav.ark", []
    with open(ark, "wb") as f:
        for i, pcm in enumerate(clips):
            f.write(f"utt{i} ".encode())
            off = f.tell()
            f.write(_riff(pcm))
            lines.append({"key": f"utt{i}", "task": "ASR" if i % 2 == 0 else "hotword", "target": f"t{i}", "path": f"{ark}:{off}",
                          "hotword": "alpha beta"})
    (tmp_path / "multitask.jsonl").write_text("\n".join(json.dumps(x) for x in lines) + "\n")
    (tmp_path / "multiprompt.jsonl").write_text(json.dumps({"task": "ASR", "prompt": "Transcribe."}) + "\n" +
                                                json.dumps({"task": "hotword", "prompt": "Use {} ."}) + "\n")
    rate, pcm = load_ark_wav(lines[1]["path"])
    assert rate == 16000 and np.array_equal(pcm, clips[1])
    tok = SimpleNamespace(encode=lambda text: [3 + (ord(c) % 50) for c in text], eos_token_id=2, pad_token_id=0)
    cfg = dict(multitask_prompt_path=str(tmp_path / "multiprompt.jsonl"), train_scp_file_path=str(tmp_path), append_info_tasks=["hotword"],
               prompt_style="USER: {}\n ASSISTANT:", pad_or_trim=False, max_audio_length=30, input_type="mel")
    ds = get_speech_dataset(cfg, tok, "train")
    assert isinstance(ds, MultiTaskDatasetRaw)
    samples = list(ds)
    assert len(samples) == 3          # the 31 s clip is dropped (speech_dataset_large.py:92-93)
    for s, line, pcm in zip(samples, lines, clips):
        assert torch.equal(s["audio"], torch.from_numpy(pcm.astype(np.float32) / 32768))
        prompt = cfg["prompt_style"].format("Transcribe." if line["task"] == "ASR" else "Use {} .")
        if line["task"] == "hotword":
            prompt = prompt.format(line["hotword"])
        alen = ((len(pcm) // 160 + 1) // 2) // 5
        pids = tok.encode(prompt)
        ref = O.make_sample(alen, pids, tok.encode(prompt + line["target"])[len(pids):], eos=2)
        assert torch.equal(s["input_ids"], ref["input_ids"].clamp(min=-1)) or torch.equal(s["input_ids"].clamp(min=0), ref["input_ids"])
        assert torch.equal(s["labels"], ref["labels"]) and s["audio_length"] == alen
    batch = ds.collator(samples)
    refb = O.collate_right_pad([O.make_sample(s["audio_length"], s["input_ids"][s["audio_length"]: s["audio_length"] + s["prompt_length"]].tolist(),
                                              s["input_ids"][s["audio_length"] + s["prompt_length"]: -1].tolist(), eos=2) for s in samples], pad_id=0)
    for k in ("labels", "attention_mask", "modality_mask"):
        assert torch.equal(batch[k].long(), refb[k].long()), k
    assert torch.equal(batch["input_ids"].clamp(min=0).masked_fill(batch["modality_mask"].bool(), 0), refb["input_ids"].masked_fill(refb["modality_mask"].bool(), 0))
    assert batch["audio"].shape == (3, 24000) and batch["audio_len"].tolist() == [16000, 4000, 24000]
    # rank 1 of 2 (one DataLoader worker each) sees the odd lines only
    ds._shard = lambda: (2, 1)
    assert [s["audio"].shape[0] for s in ds] == [4000]


def test_rank_worker_sharding_reproduces_the_reference_including_its_skip_quirk(tmp_path):
    """tests/golden/shard.json = the reference's own MultiTaskDataset.__iter__ (speech_dataset_large.py:62-156, exec'd
    unmodified by oracle/make_golden_shard.py) at world x workers = 1x1, 2x1, 2x2 over a file with too-long clips mid-file:
    the `continue` at :92-93 skips `data_index += 1`, so the dropping worker lags one line afterwards (duplicates its
    neighbour's lines).  The product must yield exactly the same utterances per (rank, worker), and the same token layout;
    `fix_shard_skip=true` is the documented deviation (disjoint shards)."""
    import json
    from slam_llm_amd.dataset import MultiTaskDatasetRaw, MultiTaskDynamicBatchDatasetRaw, get_speech_dataset
    fx = json.load(open(G.GOLD + "/shard.json"))
    ark = tmp_path / "wav.ark"
    with open(ark, "wb") as f, open(tmp_path / "multitask.jsonl", "w") as j:
        for i, sec in enumerate(fx["seconds"]):
            pcm = (np.arange(int(sec * 16000)) % 7).astype(np.int16)
            f.write(f"utt{i} ".encode())
            off = f.tell()
            f.write(_riff(pcm))
            j.write(json.dumps({"key": f"utt{i}", "task": "ASR", "target": f"text {i}", "path": f"{ark}:{off}"}) + "\n")
    (tmp_path / "multiprompt.jsonl").write_text(json.dumps({"task": "ASR", "prompt": "Transcribe."}) + "\n")

    class Tok:
        eos_token_id, pad_token_id = 2, 0

        def encode(self, text):
            return [1] + [3 + ord(c) % 50 for c in text]
    cfg = dict(multitask_prompt_path=str(tmp_path / "multiprompt.jsonl"), train_scp_file_path=str(tmp_path), append_info_tasks=[],
               prompt_style="USER: {}\n ASSISTANT:", pad_or_trim=False, max_audio_length=30, input_type="mel", inference_mode=True)
    for key, want in fx["shards"].items():
        shape, rank, wid = key.split(":")
        world, workers = (int(x) for x in shape.split("x"))
        ds = MultiTaskDatasetRaw(cfg, Tok(), "train")
        ds._shard = lambda: (world * workers, int(rank) * workers + int(wid))
        assert [s["key"] for s in ds] == want, key
    # the documented fix: disjoint shards that together cover every kept utterance exactly once
    kept = fx["shards"]["1x1:0:0"]
    seen = []
    for r in range(4):
        ds = MultiTaskDatasetRaw(dict(cfg, fix_shard_skip=True), Tok(), "train")
        ds._shard = lambda: (4, r)
        seen += [s["key"] for s in ds]
    assert sorted(seen, key=lambda k: int(k[3:])) == kept
    # training-mode token layout (prompt + answer tokenised as one string, :137-151) == the reference's samples
    ds = MultiTaskDatasetRaw(dict(cfg, inference_mode=False), Tok(), "train")
    for s, ref in zip(ds, fx["layout"]):
        assert s["input_ids"].tolist() == ref["input_ids"] and s["labels"].tolist() == ref["labels"] and s["audio_length"] == ref["audio_length"]
    # the plugin entry returns the dynamic-batch wrapper (speech_dataset_large.py:265-271): lists of samples + .collator, the
    # DataLoader contract of utils/config_utils.py:94-99 (batch_size=None, collate_fn=dataset.collator)
    dsw = get_speech_dataset(dict(cfg, inference_mode=False, train_max_frame_length=400), Tok(), "train")
    assert isinstance(dsw, MultiTaskDynamicBatchDatasetRaw)
    batches = list(torch.utils.data.DataLoader(dsw, batch_size=None, collate_fn=dsw.collator))
    assert sum(b["input_ids"].shape[0] for b in batches) == len(kept)
    assert all(b["input_ids"].shape[0] * b["input_ids"].shape[1] <= 400 or b["input_ids"].shape[0] == 1 for b in batches)


def test_packed_aware_frame_budget():
    """budget='sum' (varlen path): batches close on the SUM of real sequence lengths; same in-order greedy grouping, every
    sample exactly once, no batch above the budget unless it is a single over-long sample; 'padded' stays the reference window"""
    rng = np.random.RandomState(7)
    lens = [int(x) for x in rng.randint(40, 400, size=200)] + [5000]
    samples = [{"input_ids": torch.zeros(n, dtype=torch.int64), "idx": i} for i, n in enumerate(lens)]
    groups = list(batcher.dynamic_batches(samples, 3000, budget="sum"))
    assert [s["idx"] for g in groups for s in g] == list(range(len(lens)))
    for g in groups:
        tot = sum(len(s["input_ids"]) for s in g)
        assert tot <= 3000 or len(g) == 1
    # greedy: adding the next group's first element would have overflowed
    for g, nxt in zip(groups[:-1], groups[1:]):
        assert sum(len(s["input_ids"]) for s in g) + len(nxt[0]["input_ids"]) > 3000
    ref = list(batcher.dynamic_batches(samples, 3000))          # default = the reference's padded window
    assert [len(g) for g in ref] == [len(g) for g in O.dynamic_batches(lens, 3000)]
    n_sum, n_pad = len(groups), len(ref)
    assert n_sum <= n_pad          # counting real tokens never needs more batches than counting padded positions
    with pytest.raises(ValueError):
        list(batcher.dynamic_batches(samples, 3000, budget="mean"))
    assert batcher.frames_for_hbm() > 60000


def test_conv_stack_pitches_cover_every_valid_window():
    """HuBERT / WavLM conv layers 1-6 read layer i's [B * P_i, C] buffer as ONE overlapping-row matrix (lda = stride * C): row r of layer
    i + 1 starts at buffer row stride * r.  For that to be clip b's window t for every b, P_i = stride * P_(i+1); every valid output
    (t < T_(i+1)) must read only valid input rows (< T_i) of its own clip."""
    from slam_llm_amd.model import conv_stack_pitches
    ks, ss = (10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)
    Ts, P = conv_stack_pitches(480000, ks, ss)
    assert Ts == [95999, 47999, 23999, 11999, 5999, 2999, 1499] and P == [96000, 48000, 24000, 12000, 6000, 3000, 1500]
    for n in (16000, 11200, 48000, 400, 479999, 123457):
        Ts, P = conv_stack_pitches(n, ks, ss)
        for i in range(1, len(ks)):
            assert P[i - 1] == ss[i] * P[i] and P[i] >= Ts[i] and P[i - 1] >= Ts[i - 1]
            assert ss[i] * (Ts[i] - 1) + ks[i] - 1 <= Ts[i - 1] - 1          # the last valid window ends inside the valid rows
        assert P[-1] - Ts[-1] <= 2                                         # (the padding is a row or two per clip, not a multiple)


def test_roctx_phase_ranges_are_a_no_op_unless_enabled():
    """slam_llm_amd/trace.py (tracing row): without SLAM_ROCTX the phase() context is a shared null context (no library
    loaded); with SLAM_ROCTX=1 a child interpreter loads the roctx library of this ROCm image and pushes / pops nested ranges."""
    import subprocess
    import sys
    from slam_llm_amd import trace
    if not trace.ENABLED:
        assert trace._lib is None and trace.phase("x") is trace.phase("y")
        with trace.phase("llm_fwd"):
            pass
        trace.push("a"); trace.pop(); trace.mark("b")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from slam_llm_amd import trace\n"
            "assert trace.ENABLED and trace._lib is not None\n"
            "with trace.phase('outer'):\n"
            "    with trace.phase('inner'):\n"
            "        trace.mark('m')\n"
            "print('ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SLAM_ROCTX="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-400:]
