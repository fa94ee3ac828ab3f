"""Tensor-level wrappers over the C ABI (device memory + stream plumbing only).

torch supplies HBM allocations (`torch.empty`) and the current HIP stream; every arithmetic step is a
hand-written gfx950 kernel in libslamhip.so.  Nothing here falls back to a PyTorch op: tensors that are not
on a HIP device are rejected.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import os

import torch

from . import lib
from .lib import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SWIGLU_BWD, BF16, F32, call  # noqa: F401


class KernelTimer:
    """Optional per-launch timing with HIP events recorded on the stream the kernels are launched on (torch's
    current stream).  Enabled by bench.py over the timed region: `ops.TIMER = KernelTimer()`."""

    def __init__(self):
        self.rec = {}

    def run(self, name, work, fn, nbytes=0.0):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.rec.setdefault(name, []).append((s, e, work, nbytes))
        return out

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, lst in self.rec.items():
            ms = sum(s.elapsed_time(e) for s, e, _, _ in lst)
            out[name] = dict(launches=len(lst), total_ms=ms, avg_ms=ms / len(lst), work=sum(w for _, _, w, _ in lst),
                             bytes=sum(b for _, _, _, b in lst))
        return out


TIMER: Optional[KernelTimer] = None
TIMER_SHAPES = False   # tools: key GEMM timings by shape as well as by kernel instance


def _timed(name, work, fn, nbytes=0.0):
    """work = algorithmic FLOPs of the launch, nbytes = its algorithmic HBM bytes (operands read once + result written once)"""
    if TIMER is None:
        return fn()
    return TIMER.run(name, work, fn, nbytes)


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise lib.SlamHipError("slam_llm_amd ops need tensors resident in HBM (cuda/HIP device); got a CPU tensor")
    return ctypes.c_void_p(t.data_ptr())


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _ld(t: torch.Tensor) -> int:
    """leading dimension (elements) of a 2-D row-major view"""
    assert t.dim() == 2 and t.stride(1) == 1, "expected a 2-D tensor with unit column stride"
    return t.stride(0)


# ------------------------------------------------------------------------------------------------ GEMM
_GEMM_WS = {}   # device index -> workspace tensor of the split-K tail (kept alive for the life of the process)
GEMM_WS_BYTES = 4096 + 512 * 256 * 256 * 4   # counters + 512 slabs of 256 KiB (the auto plans use <= 256; forced sweeps up to 512): 128 MiB of 288 GB


# Round 4: the scratch buffer is registered at the first product of the process: the K-sliced forms for mid-M products (two-launch split-K
# of the 4-wave kernel) and for N <= 64 products (tall-skinny kernel) are part of the auto rule.  SLAM_GEMM_SK2=0 / SLAM_GEMM_TS=0 switch
# them off (A/B); the in-launch split-K TAIL of round 3 stays off by default (gemm_set_config(300 | 302..316)).
_SPLITK_WANTED = True
SK2_AUTO = os.environ.get("SLAM_GEMM_SK2", "1") != "0"
TS_AUTO = os.environ.get("SLAM_GEMM_TS", "1") != "0"


def _ensure_gemm_workspace(device: torch.device):
    """the split-K tail of the 4-wave GEMM needs a caller-owned scratch buffer (slam_gemm_set_workspace); registered once per
    process, on the first product after a plan was switched on (one process drives one GPU: a second device would need its own
    library state)."""
    if not _SPLITK_WANTED:
        return
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _GEMM_WS:
        return
    if _GEMM_WS:
        raise lib.SlamHipError(f"slam_llm_amd drives one GPU per process (workspace registered on cuda:{next(iter(_GEMM_WS))}, got cuda:{idx})")
    if torch.cuda.is_current_stream_capturing():
        return   # (no allocation inside a graph capture: products captured before the first eager one simply do not split)
    ws = torch.zeros(GEMM_WS_BYTES, dtype=torch.uint8, device=device)
    call("slam_gemm_set_workspace", _p(ws), GEMM_WS_BYTES)
    _GEMM_WS[idx] = ws


def gemm_nt(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, *, bias=None,
            residual=None, res_row_mod: int = 0, act: int = ACT_NONE, alpha: float = 1.0,
            out_dtype=torch.bfloat16, accumulate: bool = False, k_alg: Optional[int] = None) -> torch.Tensor:
    """out[M,N] = epi(alpha * a[M,K] @ b[N,K]^T).  a, b bf16 (row views with arbitrary ld).
    k_alg: algorithmic K (excludes zero padding) for FLOP accounting only."""
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
    M, K = a.shape
    N, K2 = b.shape
    assert K == K2, f"K mismatch {K} vs {K2}"
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    od = F32 if out.dtype == torch.float32 else BF16
    _ensure_gemm_workspace(a.device)
    if (M <= SKINNY_MAX_M and bias is None and act == ACT_NONE and alpha == 1.0 and not accumulate and res_row_mod == 0
            and _GEMM_CFG == 0):
        return gemm_skinny(a, b, out, residual=residual)  # a few rows (decode): HBM-bound weight-streaming kernel
    _timed(gemm_kernel_name(M, N, K) + (f" [{M}x{N}x{K}]" if TIMER_SHAPES else ""), 2.0 * M * N * (k_alg or K),
           lambda: call("slam_gemm_bf16_nt", _p(a), _ld(a), _p(b), _ld(b), _p(out), _ld(out), M, N, K, _p(bias),
                        _p(residual), _ld(residual) if residual is not None else 0, res_row_mod, act, alpha, od,
                        1 if accumulate else 0, _s()),
           nbytes=2.0 * (M * K + N * K) + M * N * (4 if od == F32 else 2))
    return out


SKINNY_MAX_M = 64
SKINNY_SPLITS = 0  # 0 = automatic split-K plan (tools/decode_bench.py overrides it for sweeps)


def gemm_skinny(a, b, out, residual=None, b2=None, swiglu=False):
    """out[M<=64, :] = a[M,K] @ [b; b2]^T (+ residual), or silu(gate)*up with b = [gate; up] (swiglu): every weight byte is
    read once; work = weight bytes streamed."""
    M, K = a.shape
    N, N2 = b.shape[0], (b2.shape[0] if b2 is not None else 0)
    wsb = call("slam_gemm_skinny_workspace_bytes", M, N + N2, K, SKINNY_SPLITS, 1 if swiglu else 0)
    ws = torch.empty((wsb // 4,), dtype=torch.float32, device=a.device) if wsb else None
    od = F32 if out.dtype == torch.float32 else BF16
    _timed("gemm_skinny", 2.0 * (N + N2) * K,
           lambda: call("slam_gemm_skinny_bf16_nt", _p(a), _ld(a), _p(b), _ld(b), _p(b2), _ld(b2) if b2 is not None else 0, N2,
                        _p(out), _ld(out), M, N, K, _p(residual), _ld(residual) if residual is not None else 0, od,
                        1 if swiglu else 0, _p(ws), wsb, SKINNY_SPLITS, _s()))
    return out


_GEMM_CFG = 0
_GEMM_NAMES = {1: "gemm_nt_kernel<128,128,2,2>", 11: "gemm_nt_pipe_kernel<128,128,2,2,1>", 2: "gemm_nt_kernel<256,128,4,2>", 3: "gemm_nt_kernel<128,64,2,2>",
               4: "gemm_nt_kernel<256,256,2,4>", 6: "gemm_nt_pipe_kernel<256,256,2,4,1>", 7: "gemm_nt_persist2_kernel<256,256,2,4>",
               12: "gemm_nt_w4_kernel<256,256,false,0>"}


_GEMM_BIG = {"big": 12, "shortk": 7, "small": 1}
if os.environ.get("SLAM_GEMM_SMALL"):        # sweeps: SLAM_GEMM_SMALL=1|11 (the 128x128 kernel of the auto rule)
    call("slam_gemm_set_config", 600 + int(os.environ["SLAM_GEMM_SMALL"]))
    _GEMM_BIG["small"] = int(os.environ["SLAM_GEMM_SMALL"])
if os.environ.get("SLAM_GEMM_BIG"):          # sweeps: SLAM_GEMM_BIG=12 SLAM_GEMM_BIG_SHORTK=7 python bench.py ...
    call("slam_gemm_set_config", 100 + int(os.environ["SLAM_GEMM_BIG"]))
    _GEMM_BIG["big"] = int(os.environ["SLAM_GEMM_BIG"])
if os.environ.get("SLAM_GEMM_BIG_SHORTK"):
    call("slam_gemm_set_config", 200 + int(os.environ["SLAM_GEMM_BIG_SHORTK"]))
    _GEMM_BIG["shortk"] = int(os.environ["SLAM_GEMM_BIG_SHORTK"])


if not SK2_AUTO:
    call("slam_gemm_set_config", 360)
if not TS_AUTO:
    call("slam_gemm_set_config", 370)
if os.environ.get("SLAM_GEMM_SPLITK"):       # sweeps: off | auto | <slices>
    _sk = os.environ["SLAM_GEMM_SPLITK"]
    call("slam_gemm_set_config", 301 if _sk == "off" else (300 if _sk == "auto" else 300 + int(_sk)))



def gemm_kernel_name(M: int, N: int, K: int = 1 << 30) -> str:
    """which template instance slam_gemm_bf16_nt's auto rule launches for this shape (mirrors gemm_bf16.hip)"""
    cfg = _GEMM_CFG
    if cfg in (0, 3) and N <= 64 and M <= 4096 and TS_AUTO and 256 <= K < (1 << 30):
        return "gemm_ts_kernel"
    if cfg == 0:
        tiles256 = ((M + 255) // 256) * ((N + 255) // 256)
        tiles128 = ((M + 127) // 128) * ((N + 127) // 128)
        t256 = ((tiles256 + 255) // 256) * (4.0 / 1.4)
        t128 = ((tiles128 + 511) // 512) * 2.0
        big = 6 if (_GEMM_BIG["big"] == 12 and N < 2048) else _GEMM_BIG["big"]
        cfg = 3 if N <= 64 else ((_GEMM_BIG["shortk"] if K <= 2048 else big) if t256 < t128 else _GEMM_BIG["small"])
        if N > 64 and SK2_AUTO and tiles256 <= 128 and N >= 256 and 2048 <= K < (1 << 30) and _GEMM_WS:      # mid-M: K-sliced 4-wave kernel + reduce launch
            S = min(256 // tiles256, (K // 64) // 8, 8)
            if S >= 2 and (4.0 / 1.4) / S + 4500.0 / K < 0.95 * min(t256, t128):
                return f"gemm_nt_w4_kernel<256,256,false,0> split-K x{S} + reduce"
    return _GEMM_NAMES[cfg]


def gemm_set_config(cfg: int):
    """0 = auto rule, 1 2 3 4 6 7 12 = force one kernel (tools / tests); 100 + v / 200 + v = which 256x256 kernel the auto rule uses
    for K > 2048 / K <= 2048 (sweeps)"""
    global _GEMM_CFG
    global _SPLITK_WANTED
    global SK2_AUTO, TS_AUTO
    if 360 <= cfg <= 362:
        SK2_AUTO = cfg != 360
    if cfg in (370, 371):
        TS_AUTO = cfg == 371
    if cfg in (601, 611):
        _GEMM_BIG["small"] = cfg - 600
    elif cfg >= 300:    # 300 / 301 / 302..316: split-K tail auto / off / forced slices; 400 / 401: cycle stamps off / on
        pass
    elif cfg >= 100:
        _GEMM_BIG["big" if cfg < 200 else "shortk"] = cfg % 100
    else:
        _GEMM_CFG = cfg
    call("slam_gemm_set_config", cfg)


ATTN_HEAVY_FIRST = os.environ.get("SLAM_ATTN_HEAVY", "1") != "0"     # A/B: SLAM_ATTN_HEAVY=0 python bench.py (id order of causal attention workgroups)
ATTN_QS = os.environ.get("SLAM_ATTN_QS", "1") != "0"                 # A/B: SLAM_ATTN_QS=0 (general softmax also for the LSE-less mask-free launches with a pre-scaled Q)
if not ATTN_QS:
    call("slam_attn_set_fwd_qf", 60)
if not ATTN_HEAVY_FIRST:
    call("slam_attn_set_fwd_qf", 50)
if os.environ.get("SLAM_GEMM_GROUP_M"):
    call("slam_gemm_set_group_m", int(os.environ["SLAM_GEMM_GROUP_M"]))
if os.environ.get("SLAM_GEMM_GROUP_M_RULE"):      # sweeps: "12,4,4,8" = the rule's four values (short-K many columns, long-K narrow, wide K <= 4096, other)
    call("slam_gemm_set_group_m_rule", *[int(x) for x in os.environ["SLAM_GEMM_GROUP_M_RULE"].split(",")])
_ENV_DEFAULTS = dict(_GEMM_BIG, sk2=SK2_AUTO, ts=TS_AUTO, splitk=os.environ.get("SLAM_GEMM_SPLITK", "off"))


def reset_tuning():
    """every process-global tuning knob of the library (GEMM kernel choice, split-K plans, raster group, attention variants) back to
    what this process started with (the shipped defaults, or the SLAM_GEMM_* environment of a sweep).  tests/conftest.py calls it
    before every GPU test so that a test which dies between `gemm_set_config(x)` and its own restore cannot change what the tests
    behind it measure (VERDICT r4 weak #1c)."""
    d = _ENV_DEFAULTS
    call("slam_reset_tuning")    # the library's own defaults (one place: csrc), then this process's environment overrides on top
    call("slam_set_dropout_salt", None)
    gemm_set_config(0)
    gemm_set_config(100 + d["big"])
    gemm_set_config(200 + d["shortk"])
    gemm_set_config(600 + d["small"])
    gemm_set_config(361 if d["sk2"] else 360)
    gemm_set_config(371 if d["ts"] else 370)
    gemm_set_config(301 if d["splitk"] == "off" else (300 if d["splitk"] == "auto" else 300 + int(d["splitk"])))
    gemm_set_config(320 + 4)     # split-K tail auto plan: last round <= 32 tiles ...
    gemm_set_config(340 + 2)     # ... into at most 2 slices
    gemm_set_config(400)         # cycle stamps off
    if os.environ.get("SLAM_GEMM_GROUP_M"):       # A/B: SLAM_GEMM_GROUP_M=8 restores the one-size raster of rounds 1-5 (0 / unset = per-shape rule)
        call("slam_gemm_set_group_m", int(os.environ["SLAM_GEMM_GROUP_M"]))
    if os.environ.get("SLAM_GEMM_GROUP_M_RULE"):
        call("slam_gemm_set_group_m_rule", *[int(x) for x in os.environ["SLAM_GEMM_GROUP_M_RULE"].split(",")])
    call("slam_attn_set_bwd_variant", 0)
    # auto fragments, DMA tiles, XCD-aware numbering, mask-free instantiation, transposing reads, heaviest block first, pre-scaled Q in LSE-less launches
    for knob in (0, 11, 21, 31, 41, 51 if ATTN_HEAVY_FIRST else 50, 61 if ATTN_QS else 60):
        call("slam_attn_set_fwd_qf", knob)


# ------------------------------------------------------------------------------------------------ mel
_MEL_TABLES = {}


def _hz_to_mel_slaney(f):
    import numpy as np
    f = np.asarray(f, dtype=np.float64)
    mels = 3.0 * f / 200.0
    min_log_hz, min_log_mel, logstep = 1000.0, 15.0, 27.0 / math.log(6.4)
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) * logstep, mels)


def _mel_to_hz_slaney(m):
    import numpy as np
    m = np.asarray(m, dtype=np.float64)
    f = 200.0 * m / 3.0
    min_log_hz, min_log_mel, logstep = 1000.0, 15.0, math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f)


def mel_filterbank(n_mels: int, n_fft: int = 400, sr: int = 16000):
    """librosa-slaney mel filters [n_mels, 201] (what openai-whisper ships in assets/mel_filters.npz)."""
    import numpy as np
    fft_freqs = np.linspace(0, sr / 2, n_fft // 2 + 1)
    mel_pts = np.linspace(_hz_to_mel_slaney(0.0), _hz_to_mel_slaney(sr / 2.0), n_mels + 2)
    hz_pts = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(hz_pts)
    ramps = hz_pts[:, None] - fft_freqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


def logmel_twiddle_table():
    """the folded DFT's twiddles in the layout slam_logmel_fwd streams (csrc/logmel.hip): [13 bin tiles][cos | sin][13 k-quads][64 lanes][4]
    fp32; lane = 16 * g + li holds, for k-step ks = 4 * kq + j, the value for bin 16 * tile + li at sample index n = 4 * ks + g:
    cos(2 pi bin n / 400) for n <= 200, sin(2 pi bin n / 400) for n < 200, zero beyond (and for the 7 pad bins 201..207)"""
    import numpy as np
    t, p, kq, g, li, j = np.meshgrid(np.arange(13), np.arange(2), np.arange(13), np.arange(4), np.arange(16), np.arange(4), indexing="ij")
    n = 4 * (4 * kq + j) + g
    k = 16 * t + li
    ang = 2.0 * np.pi * (k * n % 400).astype(np.float64) / 400.0
    val = np.where(p == 0, np.where(n <= 200, np.cos(ang), 0.0), np.where(n < 200, np.sin(ang), 0.0))
    val = np.where(k <= 200, val, 0.0)
    return np.ascontiguousarray(val.astype(np.float32)).reshape(13, 2, 13, 64, 4)


def _mel_tables(n_mels: int, device):
    key = (n_mels, str(device))
    if key not in _MEL_TABLES:
        import numpy as np
        n = np.arange(400, dtype=np.float64)
        window = (0.5 - 0.5 * np.cos(2 * np.pi * n / 400)).astype(np.float32)
        tw = logmel_twiddle_table()
        melT = np.ascontiguousarray(mel_filterbank(n_mels).T)  # [201, n_mels]
        _MEL_TABLES[key] = tuple(torch.from_numpy(x).to(device) for x in (window, tw, melT))
    return _MEL_TABLES[key]


def logmel(audio: torch.Tensor, n_mels: int, n_samples: int = 480000, n_valid: Optional[torch.Tensor] = None,
           per_clip: bool = False) -> torch.Tensor:
    """audio [B, N'] f32 on device -> [B, n_samples/160, n_mels] f32.
    per_clip=False: whisper.pad_or_trim(n_samples) + log_mel_spectrogram (speech_dataset.py:101-103).
    per_clip=True : pad_or_trim off -- each clip over its own n_valid samples, mel-space zero padding to n_samples/160."""
    assert audio.dtype == torch.float32 and audio.dim() == 2
    B = audio.shape[0]
    if n_valid is None and (audio.shape[1] < n_samples or per_clip):
        n_valid = torch.full((B,), min(audio.shape[1], n_samples), dtype=torch.int32, device=audio.device)
    if n_valid is not None:
        n_valid = n_valid.to(device=audio.device, dtype=torch.int32)
    window, tw, melT = _mel_tables(n_mels, audio.device)
    out = torch.empty((B, n_samples // 160, n_mels), dtype=torch.float32, device=audio.device)
    ws = torch.empty((lib.raw().slam_logmel_workspace_bytes(B) + 3) // 4, dtype=torch.int32, device=audio.device)
    call("slam_logmel_fwd", _p(audio), audio.stride(0), _p(n_valid), n_samples, _p(window), _p(tw), _p(melT),
         n_mels, _p(out), _p(ws), B, 1 if per_clip else 0, _s())
    return out


# ------------------------------------------------------------------------------------------------ conv/norm
def conv1d_k3_im2col(x: torch.Tensor, stride: int, Kp: int, n_valid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [B, Tin, C] (f32|bf16) -> [B*Tout, Kp] bf16; n_valid int32 [B]: frames >= n_valid[b] read as zero"""
    B, Tin, C = x.shape
    assert x.is_contiguous()
    Tout = (Tin + 2 - 3) // stride + 1
    out = torch.empty((B * Tout, Kp), dtype=torch.bfloat16, device=x.device)
    call("slam_conv1d_k3_im2col", _p(x), F32 if x.dtype == torch.float32 else BF16, _p(out), B, Tin, C, stride,
         Kp, _p(n_valid), _s())
    return out


def conv1d_k3_col2im(dcols: torch.Tensor, B: int, Tin: int, C: int, stride: int) -> torch.Tensor:
    """adjoint of conv1d_k3_im2col: dcols [B*Tout, >= 3C] bf16 -> dx [B, Tin, C] bf16"""
    assert dcols.dtype == torch.bfloat16 and dcols.stride(1) == 1
    dx = torch.empty((B, Tin, C), dtype=torch.bfloat16, device=dcols.device)
    call("slam_conv1d_k3_col2im", _p(dcols), _ld(dcols), _p(dx), B, Tin, C, stride, _s())
    return dx


def gather_rows(src: torch.Tensor, idx: torch.Tensor, width: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[r, :width] = src.flatten()[idx[r] * src.stride(0) : + width]; idx int32, < 0 -> zero row.  width defaults to
    src.shape[1]; a larger width spans consecutive rows of a contiguous src (the projector's k-frame windows)."""
    assert src.dtype == torch.bfloat16 and src.dim() == 2 and src.stride(1) == 1 and idx.dtype == torch.int32
    width = width or src.shape[1]
    if width > src.shape[1]:
        assert src.is_contiguous(), "multi-row windows need a contiguous source"
    n = idx.numel()
    if out is None:
        out = torch.empty((n, width), dtype=torch.bfloat16, device=src.device)
    if n:
        call("slam_gather_rows_bf16", _p(src), src.stride(0), _p(idx), _p(out), _ld(out), n, width, _s())
    return out


def layernorm(x, weight, bias, eps=1e-5, out=None, gelu=False, stats=False):
    """stats=True additionally returns (mean, rstd) fp32 [M] for slam_layernorm_bwd"""
    M, d = x.shape
    if out is None:
        out = torch.empty((M, d), dtype=torch.bfloat16, device=x.device)
    mean = rstd = None
    if stats:
        mean = torch.empty((M,), dtype=torch.float32, device=x.device)
        rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
    call("slam_layernorm_fwd", _p(x), _ld(x), _p(weight), _p(bias), _p(out), _ld(out), M, d, eps, 1 if gelu else 0,
         _p(mean), _p(rstd), _s())
    return (out, mean, rstd) if stats else out


def layernorm_bwd(x, mean, rstd, weight, dy, dgamma=None, dbeta=None, want_dx=True, accumulate=False):
    M, d = x.shape
    dx = torch.empty((M, d), dtype=torch.bfloat16, device=x.device) if want_dx else None
    call("slam_layernorm_bwd", _p(x), _ld(x), _p(mean), _p(rstd), _p(weight), _p(dy), _ld(dy), _p(dx),
         _ld(dx) if dx is not None else 0, _p(dgamma), _p(dbeta), M, d, 1 if accumulate else 0, _s())
    return dx


def gelu_fwd(z, out=None):
    M, N = z.shape
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=z.device)
    call("slam_gelu_fwd", _p(z), _ld(z), _p(out), _ld(out), M, N, _s())
    return out


def gelu_bwd(z, dy, out=None):
    M, N = z.shape
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=z.device)
    call("slam_gelu_bwd", _p(z), _ld(z), _p(dy), _ld(dy), _p(out), _ld(out), M, N, _s())
    return out


def conv1d_im2col(x2d, B, Tin, c0, C, k, stride, pad, Kp=None, Tout_limit=0, out=None):
    """x2d rows (b*Tin + t) [B*Tin, >= c0+C] (bf16|f32) -> ([B*Tout, Kp] bf16, Tout); column j*C + c"""
    Kp = Kp or round_up(k * C, 64)
    Tout = (Tin + 2 * pad - k) // stride + 1
    if Tout_limit:
        Tout = min(Tout, Tout_limit)
    if out is None:
        out = torch.empty((B * Tout, Kp), dtype=torch.bfloat16, device=x2d.device)
    call("slam_conv1d_im2col", _p(x2d), F32 if x2d.dtype == torch.float32 else BF16, _ld(x2d), c0, C, _p(out), B, Tin, k,
         stride, pad, Kp, Tout_limit, _s())
    return out, Tout


def pos_conv_supported(channels_per_group: int, taps: int) -> bool:
    return lib.raw().slam_pos_conv_supported(channels_per_group, taps) == 1


def pos_conv_pack(w_im2col: torch.Tensor, taps: int) -> torch.Tensor:
    """[G, C (co), >= taps*C] im2col-ordered weights (column j*C + ci) -> tap-major [G, taps, C (co), KP] with ci zero padded to a
    multiple of 32 (the B operand of slam_pos_conv_fwd)"""
    G, C, _ = w_im2col.shape
    KP = round_up(C, 32)
    out = torch.zeros((G, taps, C, KP), dtype=torch.bfloat16, device=w_im2col.device)
    out[..., :C] = w_im2col[:, :, : taps * C].reshape(G, C, taps, C).permute(0, 2, 1, 3)
    return out


def pos_conv_fwd(h2d: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], B: int, T: int, out: Optional[torch.Tensor] = None,
                 pre: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, pad: Optional[int] = None, act: bool = True):
    """x = residual + act(grouped_conv(h) + bias) over [B*T, d] rows, one launch.  Defaults = fairseq pos_conv + SamePad + GELU + the
    skip connection (residual = h).  pre: also receives conv + bias (training).  The adjoint: pos_conv_fwd(dconv, pos_conv_pack_adjoint(w),
    None, B, T, residual=dx, pad=taps - 1 - taps // 2, act=False)."""
    G, taps, C, _ = w_packed.shape
    if out is None:
        out = torch.empty((B * T, G * C), dtype=torch.bfloat16, device=h2d.device)
    _timed("pos_conv", 2.0 * B * T * G * C * taps * C,
           lambda: call("slam_pos_conv_fwd", _p(h2d), _ld(h2d), _p(w_packed), _p(bias), _p(out), _ld(out), _p(pre),
                        _ld(pre) if pre is not None else 0, _p(residual), _ld(residual) if residual is not None else 0, B, T, G, C, taps,
                        taps // 2 if pad is None else pad, 1 if act else 0, _s()),
           nbytes=2.0 * (2 * B * T * G * C) + 2.0 * w_packed.numel())
    return out


def pos_conv_pack_adjoint(w_im2col: torch.Tensor, taps: int) -> torch.Tensor:
    """weights of the adjoint launch: tap-reversed and channel-transposed, [G, taps, C (ci), KP (co, zero padded)]"""
    G, C, _ = w_im2col.shape
    KP = round_up(C, 32)
    out = torch.zeros((G, taps, C, KP), dtype=torch.bfloat16, device=w_im2col.device)
    w = w_im2col[:, :, : taps * C].reshape(G, C, taps, C)          # [g, co, j, ci]
    out[..., :C] = w.flip(2).permute(0, 2, 3, 1)                   # [g, j', ci, co], j' = taps - 1 - j
    return out


def conv1d_col2im(dcols: torch.Tensor, B: int, Tin: int, C: int, k: int, stride: int) -> torch.Tensor:
    """adjoint of conv1d_im2col(pad = 0): dcols [B*Tout, >= k*C] -> dx [B*Tin, C]"""
    dx = torch.empty((B * Tin, C), dtype=torch.bfloat16, device=dcols.device)
    call("slam_conv1d_col2im", _p(dcols), _ld(dcols), _p(dx), B, Tin, C, k, stride, _s())
    return dx


def rmsnorm_fwd(x, weight, eps, out=None, rstd=None):
    M, d = x.shape
    if out is None:
        out = torch.empty((M, d), dtype=torch.bfloat16, device=x.device)
    if rstd is None:
        rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
    call("slam_rmsnorm_fwd", _p(x), _ld(x), _p(weight), _p(out), _ld(out), _p(rstd), M, d, eps, _s())
    return out, rstd


def rmsnorm_bwd(x, rstd, weight, dy, dres=None, grad_scale=None, out=None):
    M, d = x.shape
    if out is None:
        out = torch.empty((M, d), dtype=torch.bfloat16, device=x.device)
    call("slam_rmsnorm_bwd", _p(x), _ld(x), _p(rstd), _p(weight), _p(dy), _ld(dy), _p(dres),
         _ld(dres) if dres is not None else 0, _p(out), _ld(out), _p(grad_scale), M, d, _s())
    return out


# ------------------------------------------------------------------------------------------------ rope / transposes
def head_rope_transpose(src2d, col0, B, T, H, D, cos=None, sin=None, inverse=False, want_t=True, Tp=None,
                        positions=None):
    """In-place RoPE on columns [col0, col0+H*D) of src2d [B*T, ld]; returns [B,H,D,Tp] transposed copy.
    positions: optional int32 [B*T] explicit rotary positions (decode / mask-derived positions of generate())."""
    Tp = Tp or round_up(T, 64)
    dst = torch.empty((B, H, D, Tp), dtype=torch.bfloat16, device=src2d.device) if want_t else None
    call("slam_head_rope_transpose", _p(src2d), _ld(src2d), col0, _p(cos), _p(sin), 1 if inverse else 0,
         _p(dst), B, T, Tp, H, D, _p(positions), _s())
    return dst


def transpose_into(old, x2d, Rp):
    """transpose into `old` when it is there with the right shape, else into a fresh tensor.  Weights DERIVED from the trainable parameters
    (transposes, re-packed taps) are refreshed after every optimizer step: they must be updated IN PLACE, because a captured training
    step (train.GraphedTrainStep) replays the addresses its kernels were launched with -- a re-allocated operand would leave every
    replay reading the copy of the capture step (round 6: the LoRA second hop read a stale A^T from the third replay on)."""
    if old is not None and tuple(old.shape) == (x2d.shape[1], Rp) and old.dtype == torch.bfloat16:
        return transpose(x2d, Rp=Rp, out=old)
    return transpose(x2d, Rp=Rp)


def transpose(x2d, Rp=None, out=None):
    R, C = x2d.shape
    Rp = Rp or round_up(R, 64)
    if out is None:
        out = torch.empty((C, Rp), dtype=torch.bfloat16, device=x2d.device)
    call("slam_transpose_bf16", _p(x2d), _ld(x2d), _p(out), _ld(out), R, C, Rp, _s())
    return out


# ------------------------------------------------------------------------------------------------ attention
def rope_inplace(src2d, col0, B, T, H, D, cos, sin, positions=None, inverse=False):
    """HF apply_rotary_pos_emb on columns [col0, col0 + H*D) of src2d [B*T, ld], in place (q and k heads are adjacent columns of the fused
    QKV buffer: one launch with H = Hq + Hkv rotates both).  No transposed copy is written: the attention kernels read their transposed
    operands from the row-major tiles (ds_read_b64_tr_b16)."""
    head_rope_transpose(src2d, col0, B, T, H, D, cos=cos, sin=sin, inverse=inverse, want_t=False, positions=positions)


def attn_needs_transposed(q2d, k2d, v2d, do2d, B, T, Tk, Hq, Hkv, D, drop=None, relpos=None) -> int:
    """bit 1: slam_attn_bwd reads Qt / Kt / dOt for this configuration (<= 64 queries, dropout, relative position bias, > 2 GiB)"""
    flags = (1 if drop else 0) | (2 if relpos else 0)
    return lib.raw().slam_attn_needs_transposed(B, T, Tk, Hq, Hkv, D, _ld(q2d), _ld(k2d), _ld(v2d), _ld(do2d) if do2d is not None else _ld(q2d), flags)


def attn_fwd(q2d, k2d, v, B, T, Hq, Hkv, D, causal, scale, key_mask=None, want_lse=True, out=None, Tk=None, seg=None,
             relpos=None, drop=None, q_prescaled=False):
    """self-attention: q/k/v rows (b*T + t), head h at column h*D.  Cross-attention: pass Tk (key / value rows b*Tk + t).
    v: the ROW-MAJOR values [B*Tk, ld] (a column slice of the fused QKV buffer) -- the kernel reads V^T with transposing LDS reads -- or,
    for the round-3 kernels (tools, A/B), the [B,Hkv,D,Tkp] transposed copy written by head_rope_transpose (a 4-D tensor).
    seg = (lo, hi) int32 [B*T]: packed sequences (B = 1), query q sees keys lo[q] <= k <= q (causal) or
    lo[q] <= k < hi[q] (bidirectional: the ragged encoder, one clip per segment).
    relpos = (gate [B,Hq,Tqp] f32, table from relpos_table(), rp_T): WavLM's gated relative position bias.
    drop = (p, seed): dropout on the attention probabilities (counter-based mask; attn_bwd with the same pair recomputes it).
    q_prescaled: q2d was produced multiplied by scale * log2(e) (QSCALE(scale): a frozen query projection with the factor folded into
    its weights -- the frozen Whisper encoder); handed to the C ABI as a negative scale."""
    Tk = Tk or T
    vt, v2d = (v, None) if v.dim() == 4 else (None, v)
    Tkp, Tqp = (vt.shape[-1] if vt is not None else round_up(Tk, 64)), round_up(T, 64)
    assert key_mask is None or key_mask.shape[-1] == Tkp
    if out is None:
        out = torch.empty((B * T, Hq * D), dtype=torch.bfloat16, device=q2d.device)
    lse = torch.empty((B, Hq, Tqp), dtype=torch.float32, device=q2d.device) if want_lse else None
    _timed("attn_fwd", 4.0 * B * Hq * T * Tk * D * (0.5 if causal else 1.0),
           lambda: call("slam_attn_fwd", _p(q2d), _ld(q2d), _p(k2d), _ld(k2d), _p(vt), _p(v2d), _ld(v2d) if v2d is not None else 0,
                        _p(out), _ld(out), _p(lse),
                        _p(key_mask), B, T, Tk, Tqp, Tkp, Hq, Hkv, D, 1 if causal else 0, (-scale if q_prescaled else scale), _p(seg[0]) if seg else None,
                        _p(seg[1]) if seg else None, _p(relpos[0]) if relpos else None,
                        (relpos[1].data_ptr() + 64 * 4) if relpos else None, relpos[2] if relpos else 0,
                        relpos[1].shape[1] if relpos else 0, float(drop[0]) if drop else 0.0,
                        (int(drop[1]) & (2 ** 64 - 1)) if drop else 0, _s()))
    return out, lse


def qscale(scale: float) -> float:
    """the factor a frozen query projection carries for attn_fwd(..., q_prescaled=True): softmax scale x log2(e)"""
    return float(scale) * 1.4426950408889634


def relpos_table(values_hr: torch.Tensor) -> torch.Tensor:
    """values_hr [H, 2T-1] f32 (bias as a function of the relative distance k - q + T - 1) -> the padded table attn_fwd's
    relpos wants: rows of stride 2T-1+128, 64 zero floats of slack on both sides"""
    H, n = values_hr.shape
    tab = torch.zeros((H, n + 128), dtype=torch.float32, device=values_hr.device)
    tab[:, 64: 64 + n] = values_hr
    return tab


def groupnorm_time_gelu(x_f32: torch.Tensor, B: int, T: int, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5, stats: bool = False):
    """x_f32 [B*T, C] fp32 (time rows of B clips) -> gelu(GroupNorm with one group per channel over each clip's T rows) as bf16
    [B*T, C]: the first conv layer of the "default" feature extractor (WavLM Base; WavLM.py:428-441).  stats=True also returns the
    [B, 2, C] (mean, rstd) block for groupnorm_time_gelu_bwd."""
    M, C = x_f32.shape
    assert M == B * T and x_f32.dtype == torch.float32
    nbytes = call("slam_groupnorm_time_workspace_bytes", B, T, C)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x_f32.device)
    y = torch.empty((M, C), dtype=torch.bfloat16, device=x_f32.device)
    call("slam_groupnorm_time_gelu", _p(x_f32), _ld(x_f32), _p(y), _ld(y), B, T, C, _p(weight), _p(bias), float(eps), _p(ws), _s())
    if stats:
        return y, ws[ws.numel() - B * 2 * C:].view(B, 2, C)
    return y


def groupnorm_time_gelu_bwd(x_f32: torch.Tensor, stats: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, dy: torch.Tensor, B: int, T: int,
                            dgamma: torch.Tensor, dbeta: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
    """backward of groupnorm_time_gelu: dy bf16 [B*T, C] -> dL/dx bf16 [B*T, C]; dgamma / dbeta fp32 [C] written or accumulated"""
    M, C = x_f32.shape
    nbytes = call("slam_groupnorm_time_workspace_bytes", B, T, C)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x_f32.device)
    dx = torch.empty((M, C), dtype=torch.bfloat16, device=x_f32.device)
    call("slam_groupnorm_time_gelu_bwd", _p(x_f32), _ld(x_f32), _p(dy), _ld(dy), _p(stats), _p(weight), _p(bias), _p(dx), _ld(dx), _p(dgamma),
         _p(dbeta), B, T, C, 1 if accumulate else 0, _p(ws), _s())
    return dx


def wavlm_gate(x2d: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, grep_a: torch.Tensor, B: int, T: int, H: int) -> torch.Tensor:
    """x2d [B*T, H*64] bf16 (the attention input), grep_linear w [8,64] / bias [8] f32, grep_a [H] f32 -> gate [B,H,Tp] f32"""
    Tp = round_up(T, 64)
    gate = torch.zeros((B, H, Tp), dtype=torch.float32, device=x2d.device)
    call("slam_wavlm_gate", _p(x2d), _ld(x2d), _p(w), _p(bias), _p(grep_a), _p(gate), B, T, H, Tp, _s())
    return gate


def weight_norm_bwd(dw: torch.Tensor, v: torch.Tensor, g: torch.Tensor, dg: torch.Tensor, dv: torch.Tensor, accumulate: bool = False):
    """dw, v, dv: fp32 [..., K] (contiguous), g, dg: fp32 with K elements -- chain rule of weight_norm(dim = last)"""
    K = v.shape[-1]
    call("slam_weight_norm_bwd", _p(dw), _p(v), _p(g), _p(dg), _p(dv), v.numel() // K, K, 1 if accumulate else 0, _s())


def relpos_bucket_grad(d_table: torch.Tensor, buckets_i32: torch.Tensor, num_buckets: int, out: torch.Tensor, accumulate: bool = False):
    """d_table: the padded table layout of relpos_table() ([H, n + 128]); out [num_buckets, H] f32"""
    H, n = d_table.shape[0], buckets_i32.numel()
    call("slam_relpos_bucket_grad", ctypes.c_void_p(d_table.data_ptr() + 64 * 4), d_table.shape[1], _p(buckets_i32), n, H, num_buckets, _p(out),
         1 if accumulate else 0, _s())


def wavlm_gate_bwd(x2d: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, grep_a: torch.Tensor, dgate: torch.Tensor, B: int, T: int, H: int):
    """backward of wavlm_gate: returns (dv [B*T*H, 8] bf16, da_term [B*T, Hp] bf16, dx [B*T, H*64] bf16) -- see slam_wavlm_gate_bwd"""
    M, Hp = B * T, round_up(H, 8)
    dv = torch.empty((M * H, 8), dtype=torch.bfloat16, device=x2d.device)
    da_term = torch.zeros((M, Hp), dtype=torch.bfloat16, device=x2d.device)
    dx = torch.empty((M, H * 64), dtype=torch.bfloat16, device=x2d.device)
    call("slam_wavlm_gate_bwd", _p(x2d), _ld(x2d), _p(w), _p(bias), _p(grep_a), _p(dgate), _p(dv), _p(da_term), _p(dx), _ld(dx), B, T, H,
         dgate.shape[-1], Hp, _s())
    return dv, da_term, dx


def attn_bwd(q2d, k2d, v2d, o2d, do2d, lse, dq2d, dk2d, dv2d, B, T, Hq, Hkv, D, causal, scale,
             key_mask=None, Tk=None, rope=None, seg=None, drop=None, relpos=None, qt=None, kt=None, dot=None):
    """rope = (cos, sin[, positions]) tables: dq/dk come out as gradients w.r.t. the pre-RoPE projections (fused
    epilogue; explicit int32 positions for packed batches).  seg = (lo, hi): packed sequences, see attn_fwd.
    qt / kt / dot: the [B,H,D,Tp] transposed copies of q / k / dO.  The shipped kernels do not read them (transposing LDS reads from the
    row-major tiles); the configurations that still run the round-2/3 kernels (slam_attn_needs_transposed: <= 64 queries, attention
    dropout, WavLM's relative position bias, tensors beyond 2 GiB, the tools' variants) get them built here when the caller passes none."""
    Tk = Tk or T
    Tqp, Tkp = round_up(T, 64), round_up(Tk, 64)
    if attn_needs_transposed(q2d, k2d, v2d, do2d, B, T, Tk, Hq, Hkv, D, drop, relpos) & 2:
        if qt is None:
            qt = head_rope_transpose(q2d, 0, B, T, Hq, D)
        if kt is None:
            kt = head_rope_transpose(k2d, 0, B, Tk, Hkv, D)
        if dot is None:
            dot = head_rope_transpose(do2d, 0, B, T, Hq, D)
    delta = torch.empty((B, Hq, Tqp), dtype=torch.float32, device=q2d.device)
    # relpos = (gate [B,Hq,Tqp] f32, table from relpos_table(), rp_T, d_gate [B,Hq,Tqp] f32 OUT, d_table (same layout as the table, ACCUMULATED)):
    # WavLM's gated relative position bias in the backward (unfrozen WavLM); dL/d(score) goes through a scratch buffer
    rp_ds = torch.empty((B, Hq, T, Tkp), dtype=torch.float32, device=q2d.device) if relpos else None
    _timed("attn_bwd", 10.0 * B * Hq * T * Tk * D * (0.5 if causal else 1.0),
           lambda: call("slam_attn_bwd", _p(q2d), _ld(q2d), _p(k2d), _ld(k2d), _p(v2d), _ld(v2d), _p(qt), _p(kt),
                        _p(o2d), _ld(o2d), _p(do2d), _ld(do2d), _p(dot), _p(lse), _p(delta), _p(key_mask), _p(dq2d),
                        _ld(dq2d), _p(dk2d), _ld(dk2d), _p(dv2d), _ld(dv2d), B, T, Tk, Tqp, Tkp, Hq, Hkv, D,
                        1 if causal else 0, scale, _p(rope[0]) if rope else None, _p(rope[1]) if rope else None,
                        _p(rope[2]) if rope and len(rope) > 2 else None, _p(seg[0]) if seg else None,
                        _p(seg[1]) if seg else None, float(drop[0]) if drop else 0.0,
                        (int(drop[1]) & (2 ** 64 - 1)) if drop else 0,
                        _p(relpos[0]) if relpos else None, (relpos[1].data_ptr() + 64 * 4) if relpos else None, relpos[2] if relpos else 0,
                        relpos[1].shape[1] if relpos else 0, _p(rp_ds), _p(relpos[3]) if relpos else None,
                        (relpos[4].data_ptr() + 64 * 4) if relpos else None, _s()))
    return delta


def attn_decode(qkv, lora_b, lora_r, cos, sin, positions, k_prompt, v_prompt, prompt_start, k_gen, v_gen, ancestors,
                gen_count_dev, gen_count, beams, Hq, Hkv, D, scale, out):
    """one decode step of attention per hypothesis: LoRA delta + RoPE + KV append of the new token, then one query per
    row against [prompt KV of its batch item | generated KV via the ancestor table]."""
    R, G = k_gen.shape[0], k_gen.shape[1]
    Tp = k_prompt.shape[1]
    _timed("attn_decode", 4.0 * R * Hkv * D * (Tp + gen_count + 1),
           lambda: call("slam_attn_decode", _p(qkv), _ld(qkv), _p(lora_b), _ld(lora_b) if lora_b is not None else 0, lora_r,
                        _p(cos), _p(sin), _p(positions), _p(k_prompt), _p(v_prompt), _p(prompt_start), _p(k_gen), _p(v_gen),
                        _p(ancestors), _p(gen_count_dev), gen_count, _p(out), _ld(out), R, beams, Tp, G, Hq, Hkv, D, scale,
                        _s()))
    return out


# ------------------------------------------------------------------------------------------------ mlp
def swiglu_fwd(gu, out=None):
    M, F2 = gu.shape
    Fd = F2 // 2
    if out is None:
        out = torch.empty((M, Fd), dtype=torch.bfloat16, device=gu.device)
    call("slam_swiglu_fwd", _p(gu), _ld(gu), _p(out), _ld(out), M, Fd, _s())
    return out


def swiglu_bwd(gu, dh, out=None, interleaved: bool = False):
    """interleaved: gu is the [gate64 | up64]-block stash of gemm_swiglu(); out is always [dgate | dup]"""
    M, F2 = gu.shape
    if out is None:
        out = torch.empty((M, F2), dtype=torch.bfloat16, device=gu.device)
    call("slam_swiglu_bwd", _p(gu), _ld(gu), _p(dh), _ld(dh), _p(out), _ld(out), M, F2 // 2, 1 if interleaved else 0, _s())
    return out


def interleave_gate_up(w: torch.Tensor) -> torch.Tensor:
    """[gate ; up] rows ([2F, K]) -> blocks of [64 gate rows ; the matching 64 up rows] (the B operand of gemm_swiglu)"""
    F2, K = w.shape
    Fd = F2 // 2
    assert Fd % 64 == 0
    return torch.stack([w[:Fd].view(Fd // 64, 64, K), w[Fd:].view(Fd // 64, 64, K)], dim=1).reshape(F2, K).contiguous()


def gemm_swiglu_supported(M: int, N: int, K: int, lda: int, ldb: int) -> bool:
    """does slam_gemm_swiglu_bf16_nt serve this shape (the auto rule runs the 4-wave kernel on it)?"""
    return _GEMM_CFG in (0, 12) and lib.raw().slam_gemm_swiglu_supported(M, N, K, lda, ldb) == 1


def gemm_swiglu(a: torch.Tensor, b_il: torch.Tensor, gu: torch.Tensor, h: torch.Tensor):
    """gu[M, 2F] (block-interleaved [gate64 | up64]) = a @ b_il^T and h[M, F] = silu(gate) * up from one launch"""
    M, K = a.shape
    N = b_il.shape[0]
    _ensure_gemm_workspace(a.device)
    _timed(_GEMM_NAMES[12] + (f" [{M}x{N}x{K}]" if TIMER_SHAPES else ""), 2.0 * M * N * K,
           lambda: call("slam_gemm_swiglu_bf16_nt", _p(a), _ld(a), _p(b_il), _ld(b_il), _p(gu), _ld(gu), _p(h), _ld(h), M, N, K, _s()),
           nbytes=2.0 * (M * K + N * K) + 2.0 * M * N + 1.0 * M * N)
    return gu, h


# ------------------------------------------------------------------------------------------------ embed / loss / optim
def embed_splice_fwd(input_ids, modality_mask_u8, embed_table, enc3d):
    B, T = input_ids.shape
    Ta, d = enc3d.shape[1], enc3d.shape[2]
    assert enc3d.stride(2) == 1 and enc3d.stride(0) == Ta * enc3d.stride(1)
    out = torch.empty((B * T, d), dtype=torch.bfloat16, device=enc3d.device)
    spans = torch.empty((B, 2), dtype=torch.int32, device=enc3d.device)
    call("slam_embed_splice_fwd", _p(input_ids), _p(modality_mask_u8), _p(embed_table), embed_table.shape[0],
         _p(enc3d), enc3d.stride(1), _p(out), _ld(out), _p(spans), B, T, Ta, d, _s())
    return out, spans


def embed_splice_bwd(spans, dx2d, B, T, Ta, d):
    out = torch.empty((B * Ta, d), dtype=torch.bfloat16, device=dx2d.device)
    call("slam_embed_splice_bwd", _p(spans), _p(dx2d), _ld(dx2d), _p(out), _ld(out), B, T, Ta, d, _s())
    return out


def ce_targets(labels, ignore_index=-100):
    B, T = labels.shape
    tgt = torch.empty((B * T,), dtype=torch.int32, device=labels.device)
    nv = torch.empty((1,), dtype=torch.int32, device=labels.device)
    call("slam_ce_targets", _p(labels), _p(tgt), _p(nv), B, T, ignore_index, _s())
    return tgt, nv


def label_rows(targets: torch.Tensor, cap: int):
    """device-side selection of the rows that carry a label (targets >= 0), in row order: returns (rows int32 [cap], tsel int32 [cap],
    inv int32 [M], count int32 [1]); positions >= count hold -1 (gather_rows turns those into zero rows, the CE kernel ignores a
    target of -1).  No host round trip: `cap` is the caller's static bound (the exact count when the host knows it)."""
    M = targets.numel()
    assert targets.dtype == torch.int32 and 0 < cap <= M
    dev = targets.device
    rows = torch.empty((cap,), dtype=torch.int32, device=dev)
    tsel = torch.empty((cap,), dtype=torch.int32, device=dev)
    inv = torch.empty((M,), dtype=torch.int32, device=dev)
    cnt = torch.empty((1,), dtype=torch.int32, device=dev)
    call("slam_label_rows", _p(targets), M, cap, _p(rows), _p(tsel), _p(inv), _p(cnt), _s())
    return rows, tsel, inv, cnt


def ce_fwd_bwd(logits2d, targets, n_valid, row_loss, row_correct, write_grad=True):
    rows, V = logits2d.shape
    call("slam_ce_fwd_bwd", _p(logits2d), _ld(logits2d), _p(targets), _p(n_valid), _p(row_loss),
         _p(row_correct), rows, V, 1 if write_grad else 0, _s())


def ce_finalize(row_loss, row_correct, n_valid):
    out = torch.empty((2,), dtype=torch.float32, device=row_loss.device)
    call("slam_ce_finalize", _p(row_loss), _p(row_correct), _p(n_valid), row_loss.numel(), _p(out), _s())
    return out


def adamw_step(p, g, m, v, p_bf16, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    call("slam_adamw_step", _p(p), _p(g), _p(m), _p(v), _p(p_bf16), p.numel(), lr, beta1, beta2, eps, wd, step,
         grad_scale, _s())


def adamw_step_dev(p, g, m, v, p_bf16, hyper, beta1, beta2, eps, wd, grad_scale=1.0):
    """adamw_step with lr, 1 - beta1^step, sqrt(1 - beta2^step) read from the device tensor `hyper` (fp32 [3]): captured steps"""
    assert hyper.dtype == torch.float32 and hyper.numel() >= 3 and hyper.is_cuda
    call("slam_adamw_step_dev", _p(p), _p(g), _p(m), _p(v), _p(p_bf16), p.numel(), _p(hyper), beta1, beta2, eps, wd, grad_scale, _s())


def adamw_hyper(lr, beta1, beta2, step, out_host: torch.Tensor):
    """fills the pinned host tensor `out_host` (fp32 [3]) with lr, 1 - beta1^step, sqrt(1 - beta2^step) as slam_adamw_step computes them"""
    assert out_host.dtype == torch.float32 and out_host.numel() >= 3 and not out_host.is_cuda
    call("slam_adamw_hyper", float(lr), float(beta1), float(beta2), int(step), ctypes.c_void_p(out_host.data_ptr()))


def set_dropout_salt(word: Optional[torch.Tensor]):
    """register (or, with None, detach) the device-resident int64 word every dropout-aware kernel XORs into its seed -- see
    include/slam_hip.h:slam_set_dropout_salt.  The caller keeps the tensor alive while it is registered."""
    if word is not None:
        assert word.dtype == torch.int64 and word.numel() == 1 and word.is_cuda
    call("slam_set_dropout_salt", _p(word) if word is not None else None)


def adamw_anyprecision_step(p, g, m_bf16, v_bf16, comp_bf16, p_bf16, lr, beta1, beta2, eps, wd, step, params_are_bf16=False):
    """AnyPrecisionAdamW step over flat buffers; the scalars are formed exactly like the reference forms them (its step counter is
    a float32 0-dim tensor, anyprecision_optimizer.py:112,137-146)"""
    t = torch.tensor(float(step))
    neg_step = -float(lr / (1 - beta1 ** t))
    dc = float((1 - beta2 ** t) ** 0.5)
    call("slam_adamw_anyprecision_step", _p(p), _p(g), _p(m_bf16), _p(v_bf16), _p(comp_bf16), _p(p_bf16), p.numel(),
         float(1 - lr * wd), 1 if wd else 0, float(beta1), float(1 - beta1), float(beta2), float(1 - beta2), dc, float(eps), neg_step,
         1 if params_are_bf16 else 0, _s())


def cast_bf16(src_f32, dst_bf16=None):
    if dst_bf16 is None:
        dst_bf16 = torch.empty(src_f32.shape, dtype=torch.bfloat16, device=src_f32.device)
    call("slam_cast_f32_to_bf16", _p(src_f32), _p(dst_bf16), src_f32.numel(), _s())
    return dst_bf16


def relu_bwd_(dh, h):
    M, N = dh.shape
    call("slam_relu_bwd", _p(dh), _ld(dh), _p(h), _ld(h), M, N, _s())
    return dh


def lora_pack_b(b_f32, scale, dst2d, dstT2d):
    """dst2d [rows, r] view (any ld) and dstT2d [r, rows] view of the fused weight / its transpose"""
    rows, r = b_f32.shape
    assert b_f32.is_contiguous()
    call("slam_lora_pack_b", _p(b_f32), scale, _p(dst2d), _ld(dst2d), _p(dstT2d), _ld(dstT2d), rows, r, _s())


def colsum(x2d, out_f32, accumulate=False):
    M, N = x2d.shape
    call("slam_colsum_bf16", _p(x2d), _ld(x2d), _p(out_f32), M, N, 1 if accumulate else 0, _s())
    return out_f32


_GRAM_WS = {}


def lora_a_fwd(x2d, a_cat, out, drop=None, pad_to=None):
    """out[M, R] = dropout(x)[M, K] @ a_cat[R, K]^T; drop = (p, seed, offset) or None (x is read once, mask in registers).
    pad_to: columns [R, pad_to) behind `out` (same rows, same leading dimension: the K-extension's padding) are zeroed by the launch."""
    M, K = x2d.shape
    R = a_cat.shape[0]
    p_, seed, off = drop if drop is not None else (0.0, 0, 0)
    _timed("lora_a_fwd", 2.0 * M * K,
           lambda: call("slam_lora_a_fwd", _p(x2d), _ld(x2d), _p(a_cat), _ld(a_cat), _p(out), _ld(out), M, R, int(pad_to or R), K, float(p_),
                        int(seed) & (2 ** 64 - 1), int(off) & (2 ** 64 - 1), _s()))
    return out


def lora_hop_dropout(du, a_t, dx, drop):
    """dx[M, K] += mask(drop) . (du[M, R] @ a_t[K, R]^T) / (1 - p) in one pass over dx (R = 32 | 64 incl. zero padding)"""
    M, R = du.shape
    K = a_t.shape[0]
    p_, seed, off = drop
    _timed("lora_hop_dropout", 2.0 * M * K * R, lambda: call("slam_lora_hop_dropout", _p(du), _ld(du), _p(a_t), _ld(a_t), _p(dx), _ld(dx), M, K, R, float(p_),
                                                             int(seed) & (2 ** 64 - 1), int(off) & (2 ** 64 - 1), _s()),
           nbytes=4.0 * M * K)
    return dx


def skinny_gram(S2d, X2d, out, out_ld_r, out_ld_c, alpha=1.0, accumulate=False, drop=None):
    """out[r*ld_r + c*ld_c] (+)= alpha * sum_m S[m,r] X'[m,c]   (LoRA dA / dB); out is an fp32 tensor (any view);
    drop = (p, seed, offset): X' = dropout(X) recomputed from the counter-based mask, else X' = X"""
    M, R = S2d.shape
    M2, C = X2d.shape
    assert M == M2
    nbytes = call("slam_skinny_gram_workspace_bytes", M, R, C)
    key = f"{S2d.device}:{torch.cuda.current_stream().cuda_stream}"     # (one per stream: the LoRA gradient products may run on a side stream)
    if key not in _GRAM_WS or _GRAM_WS[key].numel() * 4 < nbytes:   # one workspace per device, grown to the largest need
        _GRAM_WS[key] = torch.empty(nbytes // 4, dtype=torch.float32, device=S2d.device)
    p_, seed, off = drop if drop is not None else (0.0, 0, 0)
    call("slam_skinny_gram", _p(S2d), _ld(S2d), _p(X2d), _ld(X2d), _p(out), out_ld_r, out_ld_c, M, R, C, alpha,
         1 if accumulate else 0, float(p_), int(seed) & (2 ** 64 - 1), int(off) & (2 ** 64 - 1), _p(_GRAM_WS[key]), _s())
    return out


def cast_f32_(src_bf16, dst_f32, accumulate=False):
    call("slam_cast_bf16_to_f32", _p(src_bf16), _p(dst_f32), src_bf16.numel(), 1 if accumulate else 0, _s())
    return dst_f32


def add_(a2d, b2d):
    M, N = a2d.shape
    call("slam_add_bf16", _p(a2d), _ld(a2d), _p(b2d), _ld(b2d), M, N, _s())
    return a2d


def dropout(x2d, p, seed, offset, out=None, accumulate=False):
    """out (+)= mask(seed, offset) * x / (1 - p); the mask depends only on (seed, offset, element index)"""
    M, N = x2d.shape
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=x2d.device)
    call("slam_dropout_bf16", _p(x2d), _ld(x2d), _p(out), _ld(out), M, N, float(p), int(seed) & (2 ** 64 - 1),
         int(offset) & (2 ** 64 - 1), 1 if accumulate else 0, _s())
    return out
