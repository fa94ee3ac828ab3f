"""TEST INFRASTRUCTURE -- runs the REFERENCE's `MultiTaskDataset.__iter__` (src/slam_llm/datasets/speech_dataset_large.py:23-156)
over a synthetic multitask.jsonl with a too-long clip in the MIDDLE of the file, at world 2 x 2 DataLoader workers, and
records which utterances each (rank, worker) yields plus their token layout -> tests/golden/shard.json.

The class is exec'd from the reference's source text, unmodified; its third-party imports are satisfied by stubs that do
no arithmetic the fixture depends on: `kaldiio.load_mat` returns the synthetic int16 clip, `whisper.pad_or_trim` /
`log_mel_spectrogram` return arrays of the right LENGTH (only shape[0] feeds audio_length, :104-106), `dist` reports the
simulated rank/world, `torch.utils.data.get_worker_info` the simulated worker.

What this pins: the `continue` at :92-93 skips `data_index += 1` (:156), so the worker that dropped a
clip lags one line for the rest of the file.
"""
import ast
import copy
import json
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "shard.json")
SRC = "/root/reference/src/slam_llm/datasets/speech_dataset_large.py"

SECONDS = [1.0, 2.5, 0.7, 3.0, 1.2, 31.0, 0.9, 2.0, 1.5, 0.6, 2.2, 1.1, 30.5, 0.8, 1.9, 1.3, 2.8, 0.5, 1.7, 2.4]   # clips 5 and 12 exceed 30 s


class Tok:
    eos_token_id, pad_token_id = 2, 0

    def encode(self, text):
        return [1] + [3 + ord(c) % 50 for c in text]


class Cfg(dict):
    def __getattr__(self, k):
        return self[k]


def main():
    tree = ast.parse(open(SRC).read())
    clips = {f"ark:{i}": (np.arange(int(s * 16000)) % 7).astype(np.int16) for i, s in enumerate(SECONDS)}
    state = {"rank": 0, "world": 1, "worker": None}
    kaldiio = types.SimpleNamespace(load_mat=lambda p: (16000, clips[p]))

    def pad_or_trim(a, length=480000):
        return a[:length] if len(a) >= length else np.concatenate([a, np.zeros(length - len(a), dtype=a.dtype)])

    whisper = types.SimpleNamespace(pad_or_trim=pad_or_trim,
                                    log_mel_spectrogram=lambda a, n_mels=80: torch.zeros(n_mels, len(a) // 160))
    dist = types.SimpleNamespace(is_available=lambda: True, is_initialized=lambda: state["world"] > 1,
                                 get_world_size=lambda: state["world"], get_rank=lambda: state["rank"])
    ns = {"IterableDataset": torch.utils.data.IterableDataset, "torch": torch, "np": np, "json": json, "os": os, "random": random,
          "copy": copy, "kaldiio": kaldiio, "whisper": whisper, "dist": dist}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == "MultiTaskDataset":
            exec(compile(ast.Module([node], []), "speech_dataset_large.py", "exec"), ns)
    import tempfile
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "multitask.jsonl"), "w") as f:
        for i in range(len(SECONDS)):
            f.write(json.dumps({"key": f"utt{i}", "task": "ASR", "target": f"text {i}", "path": f"ark:{i}"}) + "\n")
    with open(os.path.join(tmp, "multiprompt.jsonl"), "w") as f:
        f.write(json.dumps({"task": "ASR", "prompt": "Transcribe."}) + "\n")
    cfg = Cfg(append_info_tasks=[], multitask_prompt_path=os.path.join(tmp, "multiprompt.jsonl"), train_scp_file_path=tmp,
              prompt_style="USER: {}\n ASSISTANT:", pad_or_trim=False, input_type="mel", max_audio_length=30, inference_mode=True,
              mel_size=80)
    orig = torch.utils.data.get_worker_info
    res = {"seconds": SECONDS, "source": "speech_dataset_large.py:62-156 exec'd unmodified (oracle/make_golden_shard.py)", "shards": {}}
    try:
        torch.utils.data.get_worker_info = lambda: (None if state["worker"] is None else
                                                    types.SimpleNamespace(num_workers=state["worker"][1], id=state["worker"][0]))
        for world, workers in ((1, 1), (2, 1), (2, 2)):
            for rank in range(world):
                for wid in range(workers):
                    state.update(rank=rank, world=world, worker=None if workers == 1 and world == 1 else (wid, workers))
                    ds = ns["MultiTaskDataset"](cfg, Tok(), "train")
                    res["shards"][f"{world}x{workers}:{rank}:{wid}"] = [s["key"] for s in ds]
        # token layout of the training-mode samples (prompt + answer tokenised as one string, :137-151)
        state.update(rank=0, world=1, worker=None)
        cfg["inference_mode"] = False
        ds = ns["MultiTaskDataset"](cfg, Tok(), "train")
        lay = []
        for s in list(ds)[:4]:
            lay.append({"input_ids": s["input_ids"].tolist(), "labels": s["labels"].tolist(), "audio_length": int(s["audio_length"])})
        res["layout"] = lay
    finally:
        torch.utils.data.get_worker_info = orig
    with open(OUT, "w") as f:
        json.dump(res, f)
    print("wrote", OUT)
    for k, v in res["shards"].items():
        print(k, v)


if __name__ == "__main__":
    main()
