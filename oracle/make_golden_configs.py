"""TEST INFRASTRUCTURE -- dumps the reference's recipe config dataclass DEFAULTS to tests/golden/ref_configs.json.

Runs only in the authoring container (needs /root/reference).  The two recipe config files are loaded by path, unmodified:
  examples/asr_librispeech/asr_config.py:8-130        (ModelConfig, PeftConfig, TrainConfig, DataConfig, FSDPConfig, LogConfig)
  examples/aispeech_asr/aispeech_asr_config.py:7-143  (same classes, large-scale recipe)
tests/test_plugin_boundary.py feeds these defaults (plus the `++a.b=v` overrides BASELINE.json's configs imply) to
slam_llm_amd/slam_model_hip.py:build_config / get_speech_dataset the way the reference's pipeline would
(pipeline/finetune.py:75-88: OmegaConf sub-configs read with attribute access and `.get`).
"""
import dataclasses
import importlib.util
import json
import os
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_configs.json")


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod       # dataclasses resolves string annotations through sys.modules
    spec.loader.exec_module(mod)
    return mod


def dump(mod):
    out = {}
    for cls in ("ModelConfig", "PeftConfig", "TrainConfig", "DataConfig", "FSDPConfig", "LogConfig"):
        d = dataclasses.asdict(getattr(mod, cls)())
        out[cls] = json.loads(json.dumps(d, default=str))
    return out


if __name__ == "__main__":
    res = {"asr_librispeech": dump(load(os.path.join(REF, "examples/asr_librispeech/asr_config.py"), "ref_asr_config")),
           "aispeech_asr": dump(load(os.path.join(REF, "examples/aispeech_asr/aispeech_asr_config.py"), "ref_aispeech_config")),
           "source": ["examples/asr_librispeech/asr_config.py:8-130", "examples/aispeech_asr/aispeech_asr_config.py:7-143"]}
    with open(OUT, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("wrote", OUT)
