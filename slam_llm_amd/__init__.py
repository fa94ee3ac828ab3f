"""slam_llm_amd -- MI355X-native (gfx950) hot path for SLAM-LLM's encoder -> projector -> LLM+LoRA training step.

Importing the package does not touch the GPU; `slam_llm_amd.lib` loads libslamhip.so and fails loudly
when it has not been built (no CPU fallback).
"""
__version__ = "0.1.0"
