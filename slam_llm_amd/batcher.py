"""Dynamic-frame batcher and collators of the speech recipes (host logic, no audio I/O dependencies).

Same batch MEMBERSHIP as the reference: `window_class` / `MultiTaskDynamicBatchDataset.__iter__`
(src/slam_llm/datasets/speech_dataset_large.py:244-263: greedy in-order grouping, a batch closes when
(n+1) * max_len would exceed max_frame_length; "frames" = padded LLM positions) and the two collators
(speech_dataset.py:216-291 left/right padding; speech_dataset_large.py:180-233 right padding).  Audio is NOT
converted to log-mel on the host: batches carry raw waveforms (`audio`, `audio_len`) and the GPU front end
(slam_logmel_fwd) produces `audio_mel` inside the step.  `max_frame_length` keeps its meaning; on a 288 GB part it
can be raised ~6x -- `frames_for_hbm()` gives the budget-derived cap.
"""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional

import torch

IGNORE_INDEX = -100


def window_class(elem_len: int, buffer_lens: List[int], max_frame_length: int) -> bool:
    """True -> close the current batch before adding `elem` (speech_dataset_large.py:259-263)."""
    if len(buffer_lens) == 0:
        return True
    return (len(buffer_lens) + 1) * max(elem_len, max(buffer_lens)) > max_frame_length


def dynamic_batches(samples: Iterable[dict], max_frame_length: int, budget: str = "padded") -> Iterator[List[dict]]:
    """MultiTaskDynamicBatchDataset.__iter__ (speech_dataset_large.py:244-256).
    budget "padded": the reference's window -- (n + 1) * max_len > max_frame_length closes the batch ("frames" = PADDED LLM
    positions).  budget "sum": packed-aware window for the varlen path (no pad tokens reach the LLM, `++model_config.varlen`):
    a batch closes when the SUM of its sequence lengths would exceed max_frame_length, so the budget counts real tokens;
    `frames_for_hbm()` sizes it from the stash footprint.  Same greedy in-order grouping either way (a stated extension: the
    reference has only the padded window)."""
    if budget not in ("padded", "sum"):
        raise ValueError("frame budget must be 'padded' (reference window_class) or 'sum' (packed-aware)")
    buf: List[dict] = []
    total = 0
    for elem in samples:
        n = len(elem["input_ids"])
        if budget == "padded":
            close = window_class(n, [len(e["input_ids"]) for e in buf], max_frame_length)
        else:
            close = len(buf) == 0 or total + n > max_frame_length
        if not close:
            buf.append(elem)
            total += n
        else:
            if buf:
                yield buf
            buf, total = [elem], n
    if buf:
        yield buf


def frames_for_hbm(hbm_bytes: int = 288 << 30, weights_bytes: int = 36 << 30, bytes_per_frame: int = 3_400_000) -> int:
    """largest max_frame_length whose backward stash fits (≈3.3 MB/token over 32 Llama-3-8B layers)."""
    return int((hbm_bytes * 0.85 - weights_bytes) // bytes_per_frame)


def make_sample(audio: torch.Tensor, prompt_ids: List[int], answer_ids: Optional[List[int]], eos_id: int,
                audio_length: int, example_ids: Optional[List[int]] = None) -> dict:
    """token layout [audio(-1)*audio_length, prompt, answer, eos] (speech_dataset.py:109-161).
    example_ids: the reference tokenises prompt + answer as ONE string (`tokenizer.encode(prompt + answer)`,
    speech_dataset.py:137-139) and masks the first len(prompt_ids) labels; pass that encoding here so a tokenizer that
    merges across the prompt/answer boundary yields the reference's ids (answer_ids is then ignored)."""
    if answer_ids is None and example_ids is None:  # inference_mode
        ids = torch.tensor([-1] * audio_length + list(prompt_ids), dtype=torch.int64)
        return {"input_ids": ids.clamp(min=-1), "attention_mask": ids.ge(-1), "audio": audio,
                "audio_length": audio_length, "prompt_length": len(prompt_ids)}
    text = list(example_ids) if example_ids is not None else list(prompt_ids) + list(answer_ids)
    ids = torch.tensor([-1] * audio_length + text + [eos_id], dtype=torch.int64)
    labels = ids.clone()
    labels[: audio_length + len(prompt_ids)] = IGNORE_INDEX
    return {"input_ids": ids, "labels": labels, "attention_mask": ids.ge(-1), "audio": audio,
            "audio_length": audio_length, "prompt_length": len(prompt_ids)}


def whisper_audio_length(n_samples: int, ds_rate: int = 5, pad_to_30s: bool = True) -> int:
    """((n_mel_frames + 1) // 2) // 5  (speech_dataset.py:104-105; the reference hard-codes the 5x projector rate there,
    so the dataset plugins call this with ds_rate = 5 whatever the projector's rate); 300 for a padded 30 s clip."""
    n = 480000 if pad_to_30s else n_samples
    frames = n // 160
    return ((frames + 1) // 2) // ds_rate


def raw_audio_length(n_samples: int) -> int:
    """placeholder count for raw-waveform encoders (HuBERT / WavLM): `len // 320 // 5` (speech_dataset.py:98-99,
    speech_dataset_large.py:97-98: "ad-hoc for fairseq 320x downsample", "ad-hoc for 5x fc downsample")."""
    return n_samples // 320 // 5


def _pad1(t: torch.Tensor, left: int, right: int, value) -> torch.Tensor:
    return torch.cat([torch.full((left,), value, dtype=t.dtype), t, torch.full((right,), value, dtype=t.dtype)])


def collate(samples: List[dict], pad_token_id: int, left_pad_prompt: bool, n_samples: int = 480000,
            input_type: str = "mel", pad_or_trim: bool = True) -> dict:
    """left_pad_prompt=True: SpeechDatasetJsonl.collator; False: MultiTaskDataset.collator (right padding only).
    input_type "mel": Whisper recipes -- the waveform is carried instead of the CPU log-mel (`audio` + `audio_len`, trimmed
    to n_samples as whisper.pad_or_trim would).  "raw": HuBERT / WavLM recipes -- `audio` zero padded to the longest clip
    plus the reference's float `audio_mask` (speech_dataset.py:238-244).
    "mel" batches also carry `audio_mel_post_mask` exactly as the reference's collators build it (speech_dataset.py:246-249,
    speech_dataset_large.py:198-200): [(Tmax + 1) // 2] columns, ones over each clip's own (frames + 1) // 2 encoder frames --
    frames = 3000 for every clip under pad_or_trim (whisper.pad_or_trim, speech_dataset.py:101), else the clip's own
    min(n, n_samples) // 160 with Tmax = the mel length the GPU front end produces for the batch.  Only the Q-Former branch reads
    it (slam_model.py:354-355)."""
    if left_pad_prompt:
        pl = [s["audio_length"] + s["prompt_length"] for s in samples]
        al = [len(s["input_ids"]) - p for s, p in zip(samples, pl)]
        pm, am = max(pl), max(al)
        lr = [(pm - p, am - a) for p, a in zip(pl, al)]
    else:
        tm = max(len(s["input_ids"]) for s in samples)
        lr = [(0, tm - len(s["input_ids"])) for s in samples]
    out = {
        "input_ids": torch.stack([_pad1(s["input_ids"], l, r, pad_token_id) for s, (l, r) in zip(samples, lr)]),
        "attention_mask": torch.stack([_pad1(s["attention_mask"], l, r, False) for s, (l, r) in zip(samples, lr)]),
    }
    if "labels" in samples[0]:
        out["labels"] = torch.stack([_pad1(s["labels"], l, r, IGNORE_INDEX) for s, (l, r) in zip(samples, lr)])
    mm = torch.zeros_like(out["attention_mask"])
    for i, (s, (l, _)) in enumerate(zip(samples, lr)):
        mm[i, l: l + s["audio_length"]] = True
    out["modality_mask"] = mm
    if input_type == "raw":
        alen = torch.tensor([len(s["audio"]) for s in samples], dtype=torch.int32)
        amax = int(alen.max())
        out["audio"] = torch.stack([torch.nn.functional.pad(s["audio"].float(), (0, amax - len(s["audio"]))) for s in samples])
        out["audio_mask"] = (torch.arange(amax)[None, :] < alen[:, None]).float()
        out["audio_len"] = alen
    else:
        # raw waveforms, zero padded to a common length; audio_len = true sample counts (GPU log-mel pads/trims to n_samples)
        alen = torch.tensor([min(len(s["audio"]), n_samples) for s in samples], dtype=torch.int32)
        amax = int(alen.max())
        out["audio"] = torch.stack([torch.nn.functional.pad(s["audio"][:amax].float(), (0, amax - min(len(s["audio"]), amax)))
                                    for s in samples])
        out["audio_len"] = alen
        if pad_or_trim:
            frames, tmax = [n_samples // 160] * len(samples), n_samples // 160
        else:
            frames = [int(n) // 160 for n in alen]
            tmax = min(n_samples, (amax + 159) // 160 * 160) // 160      # SlamHipModel.forward: per-clip mel zero padded to the batch max
        pmask = torch.zeros(len(samples), (tmax + 1) // 2)
        for i, fr in enumerate(frames):
            pmask[i, : (fr + 1) // 2] = 1
        out["audio_mel_post_mask"] = pmask
    out["audio_len_list"] = [int(x) for x in out["audio_len"]]   # host copy: survives the train loop's tensor-only .to(device)
    if "key" in samples[0]:  # inference-mode batches carry the utterance ids / references (speech_dataset.py:259-273)
        out["keys"] = [s.get("key") for s in samples]
        out["targets"] = [s.get("target") for s in samples]
    return out
