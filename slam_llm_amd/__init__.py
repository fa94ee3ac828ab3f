"""slam_llm_amd -- MI355X-native (gfx950) hot path for SLAM-LLM's encoder -> projector -> LLM+LoRA training step.

Importing the package does not touch the GPU; `slam_llm_amd.lib` loads libslamhip.so and fails loudly
when it has not been built (no CPU fallback).
"""
import os as _os

# HIP runtime configuration (INTEGRATION.md "Runtime environment"): kernel arguments go straight to device memory -- 0.6 % (C3) to 4.7 % (C4) of a training step on
# MI355X (profiles/r06_runtime_env.md).  Only effective when this runs before the HIP runtime initialises (import this package, or export the variable, before
# the first torch.cuda call); an exported value wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__version__ = "0.1.0"
