"""Dataset plugins: JSONL (asr_librispeech) and kaldi-ark multitask (aispeech_asr) speech datasets feeding RAW
waveforms to the GPU log-mel front end.

Mirrors `SpeechDatasetJsonl` (src/slam_llm/datasets/speech_dataset.py:17-298): same JSONL schema
({"key","source","target"}, examples/asr_librispeech/README.md:17-22), same prompt template and token layout, same
collator padding rules -- but `__getitem__` does NOT run whisper.log_mel_spectrogram on the CPU worker
(speech_dataset.py:101-103); the collated batch carries `audio` [B, N] + `audio_len`, and SlamHipModel.forward runs
slam_logmel_fwd on the device.  With `dataset_config.train_max_frame_length` set, `dynamic_batches()` reproduces
the reference's dynamic-frame batching (speech_dataset_large.py:244-263).  Audio decoding uses the stdlib `wave`
module (16 kHz PCM wav); other containers need the reference's ffmpeg path and are out of scope here.
"""
from __future__ import annotations

import json
import os
import random
import struct
import wave

import numpy as np
import torch

from .batcher import collate, dynamic_batches, make_sample, raw_audio_length, whisper_audio_length


def load_wav_16k(path: str) -> torch.Tensor:
    with wave.open(path, "rb") as w:
        if w.getframerate() != 16000:
            raise ValueError(f"{path}: expected 16 kHz audio, got {w.getframerate()}")
        n, ch, sw = w.getnframes(), w.getnchannels(), w.getsampwidth()
        raw = w.readframes(n)
    if sw != 2:
        raise ValueError(f"{path}: expected 16-bit PCM")
    a = np.frombuffer(raw, dtype=np.int16).astype(np.float32) / 32768.0
    if ch > 1:
        a = a.reshape(-1, ch).mean(axis=1)
    return torch.from_numpy(a)


def _cfg_get(dataset_config):
    """`.get(key, default)` over the reference's OmegaConf dataset_config, a plain dict, or a dataclass instance"""
    if hasattr(dataset_config, "get"):
        return dataset_config.get
    return lambda k, d=None: getattr(dataset_config, k, d)


def _input_type(g):
    it = g("input_type", None)
    if it not in ("raw", "mel"):   # speech_dataset.py:53: assert self.input_type in ["raw", "mel"]
        raise ValueError("dataset_config.input_type must be one of [raw, mel] (raw: HuBERT / WavLM waveforms, mel: Whisper)")
    return it


def _prepare_audio(audio: torch.Tensor, input_type: str, normalize: bool, fix_length_audio: int, pad_to_30s: bool):
    """(audio, audio_length): the placeholder arithmetic of speech_dataset.py:94-108 / speech_dataset_large.py:94-108."""
    if input_type == "raw":
        if normalize:
            audio = torch.nn.functional.layer_norm(audio, audio.shape)
        alen = raw_audio_length(len(audio))
    else:
        alen = whisper_audio_length(len(audio), 5, pad_to_30s=pad_to_30s)
    if fix_length_audio > 0:
        alen = fix_length_audio
    return audio, alen


class SpeechDatasetJsonlRaw(torch.utils.data.Dataset):
    def __init__(self, dataset_config, tokenizer=None, split="train"):
        super().__init__()
        g = _cfg_get(dataset_config)
        self.tokenizer = tokenizer
        self.prompt = g("prompt", None) or ("Transcribe speech to text. Output the transcription directly without "
                                            "redundant content. Ensure that the output is not duplicated. ")
        self.prompt_template = "USER: {}\n ASSISTANT:"
        self.fix_length_audio = g("fix_length_audio", -1)
        self.inference_mode = g("inference_mode", False)
        self.normalize = bool(g("normalize", False))
        self.input_type = _input_type(g)
        self.left_pad = g("left_pad_prompt", True)
        self.max_frame_length = g("train_max_frame_length" if split == "train" else "eval_max_frame_length", None)
        path = g("train_data_path") if split == "train" else g("val_data_path")
        self.data_list = []
        with open(path, encoding="utf-8") as fin:
            for line in fin:
                if line.strip():
                    self.data_list.append(json.loads(line))

    def __len__(self):
        return len(self.data_list)

    def __getitem__(self, index):
        d = self.data_list[index]
        audio, alen = _prepare_audio(load_wav_16k(d["source"]), self.input_type, self.normalize, self.fix_length_audio, True)
        prompt = self.prompt_template.format(self.prompt)
        prompt_ids = self.tokenizer.encode(prompt)
        if self.inference_mode:
            s = make_sample(audio, prompt_ids, None, self.tokenizer.eos_token_id, alen)
            s.update(key=d.get("key"), target=d.get("target"))
            return s
        # speech_dataset.py:136-139: prompt + answer tokenised as ONE string, the first len(prompt_ids) labels masked
        example_ids = self.tokenizer.encode(prompt + "{}".format(d.get("target", None)))
        return make_sample(audio, prompt_ids, None, self.tokenizer.eos_token_id, alen, example_ids=example_ids)

    def collator(self, samples):
        return collate(samples, self.tokenizer.pad_token_id, self.left_pad, input_type=self.input_type)

    def dynamic_batch_iter(self):
        """in-order dynamic-frame batches (already collated)"""
        for group in dynamic_batches((self[i] for i in range(len(self))), int(self.max_frame_length)):
            yield self.collator(group)


def load_ark_wav(spec: str):
    """`path:offset` entry of a kaldi wav.ark / scp line (what `kaldiio.load_mat` resolves at
    src/slam_llm/datasets/speech_dataset_large.py:90-91): a RIFF/WAVE file embedded at byte `offset`.
    Returns (sample_rate, int16 ndarray, mono).  A plain path (no offset) is read from byte 0."""
    path, _, off = spec.rpartition(":")
    if not path or not off.isdigit():
        path, off = spec, "0"
    with open(path, "rb") as f:
        f.seek(int(off))
        riff, _size, wave_id = struct.unpack("<4sI4s", f.read(12))
        if riff != b"RIFF" or wave_id != b"WAVE":
            raise ValueError(f"{spec}: not a RIFF/WAVE entry (kaldi feature matrices are not audio)")
        fmt = None
        while True:
            hdr = f.read(8)
            if len(hdr) < 8:
                raise ValueError(f"{spec}: no data chunk")
            cid, csize = struct.unpack("<4sI", hdr)
            if cid == b"fmt ":
                fmt = struct.unpack("<HHIIHH", f.read(16))
                f.seek(csize - 16, 1)
            elif cid == b"data":
                if fmt is None or fmt[0] != 1 or fmt[5] != 16:
                    raise ValueError(f"{spec}: expected 16-bit PCM")
                if csize in (0, 0xFFFFFFFF):        # streamed ark entries leave the size open
                    raw = f.read()
                else:
                    raw = f.read(csize)
                a = np.frombuffer(raw[: len(raw) // 2 * 2], dtype=np.int16)
                if fmt[1] > 1:
                    a = a.reshape(-1, fmt[1]).mean(axis=1).astype(np.int16)
                return fmt[2], a
            else:
                f.seek(csize + (csize & 1), 1)


class MultiTaskDatasetRaw(torch.utils.data.IterableDataset):
    """`MultiTaskDataset` (src/slam_llm/datasets/speech_dataset_large.py:23-233) over raw waveforms: same files
    (`<split>_scp_file_path/multitask.jsonl` with {"key","task","target","path"}, `multitask_prompt_path` with
    {"task","prompt"}), same rank x worker line sharding (:80-86), `max_audio_length` filter (:92-93), random prompt
    per sample (:113), `append_info_tasks` (:115-116), token layout and RIGHT-padding collator (:180-233) -- but the
    log-mel is not computed here (:102-104): the batch carries `audio` + `audio_len` for slam_logmel_fwd on the device.
    `audio_length` reproduces the reference's mel-frame arithmetic for `pad_or_trim` on or off.

    Sharding quirk, reproduced by default because index work must match the reference bit for bit: the reference's
    `continue` for a clip longer than max_audio_length skips its `data_index += 1` (:92-93 vs :156), so the worker that
    dropped a clip stays one line behind for the rest of the file -- from then on it yields the lines of the NEXT worker
    rank (which that worker yields too) and never its own.  `dataset_config.fix_shard_skip=true` advances the index on
    dropped clips instead (disjoint shards; a conscious deviation)."""

    def __init__(self, dataset_config, tokenizer=None, split="train"):
        super().__init__()
        g = _cfg_get(dataset_config)
        self.prompts = {}
        with open(g("multitask_prompt_path")) as f:
            for line in f:
                if line.strip():
                    item = json.loads(line)
                    self.prompts.setdefault(item["task"], []).append(item["prompt"])
        key = {"train": "train_scp_file_path", "val": "dev_scp_file_path", "test": "test_scp_file_path"}.get(split)
        if key is None:
            raise ValueError("split must be train val test")
        self.data_path = g(key)
        self.append_info_tasks = g("append_info_tasks", None) or []
        self.prompt_style = g("prompt_style", "{}")
        self.tokenizer = tokenizer
        self.pad_or_trim = g("pad_or_trim", False)
        self.fix_length_audio = g("fix_length_audio", -1)
        self.inference_mode = g("inference_mode", False)
        self.normalize = bool(g("normalize", False))
        self.input_type = _input_type(g)
        self.max_audio_length = g("max_audio_length", 30)
        self.audio_sample_rate = g("audio_sample_rate", 16000)
        self.fix_shard_skip = bool(g("fix_shard_skip", False))
        self.max_frame_length = g("train_max_frame_length" if split == "train" else "eval_max_frame_length", None)

    def _shard(self):
        info = torch.utils.data.get_worker_info()
        nw, wid = (info.num_workers, info.id) if info is not None else (1, 0)
        import torch.distributed as dist
        ws, rk = (dist.get_world_size(), dist.get_rank()) if dist.is_available() and dist.is_initialized() else (1, 0)
        return nw * ws, rk * nw + wid

    def __iter__(self):
        total, mine = self._shard()
        data_index = 0
        with open(os.path.join(self.data_path, "multitask.jsonl")) as f:
            for line in f:
                if data_index % total != mine:
                    data_index += 1
                    continue
                item = json.loads(line)
                rate, pcm = load_ark_wav(item["path"])
                if rate != self.audio_sample_rate:
                    raise ValueError(f"{item['path']}: expected {self.audio_sample_rate} Hz audio, got {rate}")
                audio = torch.from_numpy(pcm.astype(np.float32) / 32768)
                if len(audio) / self.audio_sample_rate > self.max_audio_length:
                    if self.fix_shard_skip:
                        data_index += 1
                    continue   # reference: the index does NOT advance here (class docstring)
                data_index += 1
                audio, alen = _prepare_audio(audio, self.input_type, self.normalize, self.fix_length_audio, bool(self.pad_or_trim))
                prompt = self.prompt_style.format(random.choice(self.prompts[item["task"]]))
                if item["task"] in self.append_info_tasks:
                    prompt = prompt.format(item[item["task"]])
                prompt_ids = self.tokenizer.encode(prompt)
                if self.inference_mode:
                    s = make_sample(audio, prompt_ids, None, self.tokenizer.eos_token_id, alen)
                    s.update(key=item["key"], target=item["target"])
                else:   # :137-139: prompt + answer tokenised as one string
                    s = make_sample(audio, prompt_ids, None, self.tokenizer.eos_token_id, alen,
                                    example_ids=self.tokenizer.encode(prompt + "{}".format(item["target"])))
                yield s

    def collator(self, samples):
        return collate(samples, self.tokenizer.pad_token_id, left_pad_prompt=False, input_type=self.input_type,
                       pad_or_trim=bool(self.pad_or_trim))

    def dynamic_batch_iter(self):
        """in-order batches under max_frame_length, already collated"""
        for group in dynamic_batches(iter(self), int(self.max_frame_length)):
            yield self.collator(group)


class MultiTaskDynamicBatchDatasetRaw(torch.utils.data.IterableDataset):
    """`MultiTaskDynamicBatchDataset` (speech_dataset_large.py:235-256): yields LISTS of samples grouped by the dynamic-frame
    window; the reference's DataLoader is built with `batch_size=None, collate_fn=dataset.collator`
    (utils/config_utils.py:94-99), so this object exposes the wrapped dataset's collator."""

    def __init__(self, dataset: MultiTaskDatasetRaw, max_frame_length: int, budget: str = "padded"):
        super().__init__()
        self.dp = dataset
        self.max_frame_length = int(max_frame_length)
        self.budget = budget   # dataset_config.frame_budget: "padded" (reference window) | "sum" (packed-aware, varlen path)
        self.collator = dataset.collator

    def __iter__(self):
        return dynamic_batches(iter(self.dp), self.max_frame_length, self.budget)


def get_speech_dataset(dataset_config, tokenizer, split):
    """plugin entry (dataset_config.file = ".../slam_model_hip.py:get_speech_dataset"): kaldi-ark multitask layout when
    the config names `multitask_prompt_path` (aispeech_asr recipes; returns the dynamic-batch wrapper like
    speech_dataset_large.py:265-271), JSONL otherwise (asr_librispeech recipes, speech_dataset.py:295-298)."""
    g = _cfg_get(dataset_config)
    if g("multitask_prompt_path", None):
        ds = MultiTaskDatasetRaw(dataset_config, tokenizer, split)
        mfl = g("train_max_frame_length" if split == "train" else "eval_max_frame_length", None)
        return MultiTaskDynamicBatchDatasetRaw(ds, mfl, g("frame_budget", "padded")) if mfl else ds
    return SpeechDatasetJsonlRaw(dataset_config, tokenizer, split)
