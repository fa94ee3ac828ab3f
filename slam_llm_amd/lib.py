"""ctypes binding of libslamhip.so (C ABI declared in include/slam_hip.h).

This is the reference-side stub a SLAM-LLM maintainer would add: plain pointers and
sizes in, an int return code out, ``RuntimeError(slam_last_error())`` on failure -- mirroring the reference's
"raise a Python exception" convention (src/slam_llm/utils/model_utils.py:17-23).  There is NO CPU fallback:
if the shared library is missing or a symbol is absent, importing this module fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

# torch must be loaded first: it brings the process-wide HIP runtime (libamdhip64) that owns the device context,
# the allocator and the streams this library launches on; loading libslamhip.so before torch would bind it to a
# second, separate runtime copy ("no ROCm-capable device is detected" at the first launch).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLAM_HIP_LIB: another build of the same library (tools: A/B of two kernel versions inside one gpurun call); never a fallback
LIB_PATH = os.environ.get("SLAM_HIP_LIB") or os.path.join(_HERE, "libslamhip.so")

ABI_VERSION = 2      # what SIGNATURES below was written for; _load() refuses any other library (include/slam_hip.h, csrc/capi_core.hip)
BF16, F32 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU, ACT_SWIGLU_BWD = 0, 1, 2, 3

P, I64, I32, F, U64 = c_void_p, c_int64, c_int, c_float, ctypes.c_uint64

# name -> argtypes; every symbol of include/slam_hip.h (tests/test_capi_symbols.py checks the two agree)
SIGNATURES = {
    "slam_reset_tuning": [],
    "slam_set_dropout_salt": [P],
    "slam_label_rows": [P, I64, I64, P, P, P, P, P],
    "slam_adamw_hyper": [F, F, F, I64, P],
    "slam_adamw_step_dev": [P, P, P, P, P, I64, P, F, F, F, F, F, P],
    "slam_logmel_workspace_bytes": [I64],
    "slam_logmel_fwd": [P, I64, P, I64, P, P, P, I64, P, P, I64, I32, P],
    "slam_gemm_bf16_nt": [P, I64, P, I64, P, I64, I64, I64, I64, P, P, I64, I64, I32, F, I32, I32, P],
    "slam_gemm_set_config": [I32],
    "slam_gemm_set_group_m": [I32],
    "slam_gemm_set_group_m_rule": [I32, I32, I32, I32],
    "slam_gemm_set_workspace": [P, I64],
    "slam_gemm_debug_clock": [P],
    "slam_attn_set_fwd_qf": [I32],
    "slam_attn_set_bwd_variant": [I32],
    "slam_attn_debug_clock": [P],
    "slam_conv1d_k3_im2col": [P, I32, P, I64, I64, I64, I64, I64, P, P],
    "slam_conv1d_k3_col2im": [P, I64, P, I64, I64, I64, I64, P],
    "slam_gather_rows_bf16": [P, I64, P, P, I64, I64, I64, P],
    "slam_conv1d_im2col": [P, I32, I64, I64, I64, P, I64, I64, I64, I64, I64, I64, I64, P],
    "slam_layernorm_fwd": [P, I64, P, P, P, I64, I64, I64, F, I32, P, P, P],
    "slam_layernorm_bwd": [P, I64, P, P, P, P, I64, P, I64, P, P, I64, I64, I32, P],
    "slam_gelu_fwd": [P, I64, P, I64, I64, I64, P],
    "slam_gelu_bwd": [P, I64, P, I64, P, I64, I64, I64, P],
    "slam_rmsnorm_fwd": [P, I64, P, P, I64, P, I64, I64, F, P],
    "slam_rmsnorm_bwd": [P, I64, P, P, P, I64, P, I64, P, I64, P, I64, I64, P],
    "slam_head_rope_transpose": [P, I64, I64, P, P, I32, P, I64, I64, I64, I64, I64, P, P],
    "slam_transpose_bf16": [P, I64, P, I64, I64, I64, I64, P],
    "slam_attn_needs_transposed": [I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I32],
    "slam_attn_fwd": [P, I64, P, I64, P, P, I64, P, I64, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I32, F, P, P, P, P, I64, I64, F, U64, P],
    "slam_wavlm_gate": [P, I64, P, P, P, P, I64, I64, I64, I64, P],
    "slam_weight_norm_bwd": [P, P, P, P, P, I64, I64, I32, P],
    "slam_relpos_bucket_grad": [P, I64, P, I64, I64, I64, P, I32, P],
    "slam_wavlm_gate_bwd": [P, I64, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, P],
    "slam_groupnorm_time_workspace_bytes": [I64, I64, I64],
    "slam_groupnorm_time_gelu": [P, I64, P, I64, I64, I64, I64, P, P, F, P, P],
    "slam_groupnorm_time_gelu_bwd": [P, I64, P, I64, P, P, P, P, I64, P, P, I64, I64, I64, I32, P, P],
    "slam_attn_bwd": [P, I64, P, I64, P, I64, P, P, P, I64, P, I64, P, P, P, P, P, I64, P, I64, P, I64,
                      I64, I64, I64, I64, I64, I64, I64, I64, I32, F, P, P, P, P, P, F, U64, P, P, I64, I64, P, P, P, P],
    "slam_pos_conv_supported": [I64, I64],
    "slam_pos_conv_fwd": [P, I64, P, P, P, I64, P, I64, P, I64, I64, I64, I64, I64, I64, I64, I32, P],
    "slam_conv1d_col2im": [P, I64, P, I64, I64, I64, I64, I64, P],
    "slam_swiglu_fwd": [P, I64, P, I64, I64, I64, P],
    "slam_swiglu_bwd": [P, I64, P, I64, P, I64, I64, I64, I32, P],
    "slam_gemm_swiglu_supported": [I64, I64, I64, I64, I64],
    "slam_gemm_swiglu_bf16_nt": [P, I64, P, I64, P, I64, P, I64, I64, I64, I64, P],
    "slam_relu_bwd": [P, I64, P, I64, I64, I64, P],
    "slam_colsum_bf16": [P, I64, P, I64, I64, I32, P],
    "slam_skinny_gram_workspace_bytes": [I64, I64, I64],
    "slam_skinny_gram": [P, I64, P, I64, P, I64, I64, I64, I64, I64, F, I32, F, U64, U64, P, P],
    "slam_lora_a_fwd": [P, I64, P, I64, P, I64, I64, I64, I64, I64, F, U64, U64, P],
    "slam_lora_hop_dropout": [P, I64, P, I64, P, I64, I64, I64, I64, F, U64, U64, P],
    "slam_lora_pack_b": [P, F, P, I64, P, I64, I64, I64, P],
    "slam_gemm_skinny_workspace_bytes": [I64, I64, I64, I64, I32],
    "slam_gemm_skinny_bf16_nt": [P, I64, P, I64, P, I64, I64, P, I64, I64, I64, I64, P, I64, I32, I32, P, I64, I64, P],
    "slam_attn_decode": [P, I64, P, I64, I64, P, P, P, P, P, P, P, P, P, P, I64, P, I64, I64, I64, I64, I64, I64, I64,
                         I64, F, P],
    "slam_embed_splice_fwd": [P, P, P, I64, P, I64, P, I64, P, I64, I64, I64, I64, P],
    "slam_embed_splice_bwd": [P, P, I64, P, I64, I64, I64, I64, I64, P],
    "slam_ce_targets": [P, P, P, I64, I64, I64, P],
    "slam_ce_fwd_bwd": [P, I64, P, P, P, P, I64, I64, I32, P],
    "slam_ce_finalize": [P, P, P, I64, P, P],
    "slam_adamw_step": [P, P, P, P, P, I64, F, F, F, F, F, I64, F, P],
    "slam_adamw_anyprecision_step": [P, P, P, P, P, P, I64, F, I32, F, F, F, F, F, F, F, I32, P],
    "slam_cast_f32_to_bf16": [P, P, I64, P],
    "slam_cast_bf16_to_f32": [P, P, I64, I32, P],
    "slam_dropout_bf16": [P, I64, P, I64, I64, I64, F, ctypes.c_uint64, ctypes.c_uint64, I32, P],
    "slam_add_bf16": [P, I64, P, I64, I64, I64, P],
}


class SlamHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the SLAM hot path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.slam_last_error.restype = c_char_p
    lib.slam_last_error.argtypes = []
    lib.slam_abi_version.restype = c_int
    lib.slam_target_arch.restype = c_char_p
    got = lib.slam_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} reports C-ABI version {got}; this binding was written for version {ABI_VERSION} (argument lists differ "
                          "between versions: rebuild the library from this tree's csrc/)")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so and the header ever diverge
        fn.argtypes = argtypes
        fn.restype = c_int64 if name.endswith("_workspace_bytes") else c_int
    return lib


_lib = _load()


def raw():
    """The ctypes CDLL handle (for symbol checks)."""
    return _lib


def last_error() -> str:
    return (_lib.slam_last_error() or b"").decode()


def call(name: str, *args) -> int:
    """Invoke a C-ABI entry point; raise SlamHipError on a non-zero return code."""
    rc = getattr(_lib, name)(*args)
    if name.endswith("_workspace_bytes"):
        return rc
    if rc != 0:
        raise SlamHipError(f"{name} failed (rc={rc}): {last_error()}")
    return rc
