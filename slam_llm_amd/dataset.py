"""Dataset plugin: JSONL speech dataset feeding RAW waveforms to the GPU log-mel front end.

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
import wave

import numpy as np
import torch

from .batcher import collate, dynamic_batches, make_sample, whisper_audio_length


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


class SpeechDatasetJsonlRaw(torch.utils.data.Dataset):
    def __init__(self, dataset_config, tokenizer=None, split="train"):
        super().__init__()
        g = dataset_config.get
        self.tokenizer = tokenizer
        self.prompt = g("prompt", None) or ("Transcribe speech to text. Output the transcription directly without "
                                            "redundant content. Ensure that the output is not duplicated. ")
        self.prompt_template = "USER: {}\n ASSISTANT:"
        self.fix_length_audio = g("fix_length_audio", -1)
        self.inference_mode = g("inference_mode", False)
        self.ds_rate = g("encoder_projector_ds_rate", 5)
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
        audio = load_wav_16k(d["source"])
        alen = self.fix_length_audio if self.fix_length_audio > 0 else whisper_audio_length(len(audio), self.ds_rate)
        prompt_ids = self.tokenizer.encode(self.prompt_template.format(self.prompt))
        if self.inference_mode:
            s = make_sample(audio, prompt_ids, None, self.tokenizer.eos_token_id, alen)
            s.update(key=d.get("key"), target=d.get("target"))
            return s
        full = self.tokenizer.encode(self.prompt_template.format(self.prompt) + str(d.get("target", "")))
        return make_sample(audio, prompt_ids, full[len(prompt_ids):], self.tokenizer.eos_token_id, alen)

    def collator(self, samples):
        return collate(samples, self.tokenizer.pad_token_id, self.left_pad)

    def dynamic_batch_iter(self):
        """in-order dynamic-frame batches (already collated)"""
        for group in dynamic_batches((self[i] for i in range(len(self))), int(self.max_frame_length)):
            yield self.collator(group)


def get_speech_dataset(dataset_config, tokenizer, split):
    return SpeechDatasetJsonlRaw(dataset_config, tokenizer, split)
