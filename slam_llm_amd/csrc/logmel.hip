// Whisper log-mel front end on the GPU (gfx950), fp32 end to end.
//
// Replaces the per-sample CPU DataLoader work whisper.pad_or_trim + whisper.log_mel_spectrogram called at
// src/slam_llm/datasets/speech_dataset.py:101-103 and speech_dataset_large.py:102-104 (third-party
// openai-whisper; algorithm restated from the HF twin transformers/models/whisper/
// feature_extraction_whisper.py:135-168): zero-pad/trim to N samples, STFT(n_fft 400, hop 160, periodic
// Hann, center=True with reflect padding), drop the last frame, |.|^2, slaney mel filterbank (201 -> n_mels),
// log10(clamp 1e-10), per-clip floor at (max - 8) computed over ALL frames incl. the silent tail,
// (x + 4) / 4.  Output layout [B, n_frames, n_mels] = the batch dict's `audio_mel`.
//
// Kernel 1: one workgroup = 32 frames of one clip.  The windowed frames live in LDS ([32][401] fp32); the
// 400-point real DFT is a [32 x 400] . [400 x 416] product on the exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32: bitwise an fmaf chain), cos columns 0..207 / sin columns 208..415 of a
// host-precomputed twiddle table (L2 resident, 650 KB).  Power goes back to LDS, the mel projection is
// a short VALU dot against the transposed filterbank, and the clip maximum is reduced with an ordered-int
// atomicMax.  Kernel 2 applies the floor and the affine map in place.
#include "common.h"
#include <limits.h>
#include <algorithm>

namespace {

constexpr int NFFT = 400, HOP = 160, NBIN = 201, NBINP = 208, TWC = 416, FR = 32, XLD = 401, PLD = 209;

__device__ __forceinline__ int f2ord(float f) {
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ void logmel_init_kernel(int* clipmax, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) clipmax[i] = INT_MIN;
}

__global__ __launch_bounds__(256) void logmel_stft_kernel(const float* __restrict__ audio, int64_t ld_audio,
                                                          const int* __restrict__ n_valid, int N,
                                                          const float* __restrict__ window,
                                                          const float* __restrict__ twiddle,
                                                          const float* __restrict__ melT, int n_mels,
                                                          float* __restrict__ out, int n_frames,
                                                          int* __restrict__ clipmax, int per_clip) {
  __shared__ float lds[FR * XLD];  // frames [32][401]; later re-used as power [32][209]
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int b = blockIdx.y, f0 = blockIdx.x * FR;
  const float* x = audio + (int64_t)b * ld_audio;
  const int nv = n_valid ? min(n_valid[b], N) : N;
  // per_clip: the clip is NOT padded to N first (pad_or_trim off, speech_dataset_large.py:102-104): its own length is
  // the STFT length (reflection about its own end), it owns nv/160 frames and the rest of the row is mel-space zero
  // padding added by the collator (speech_dataset_large.py:194-197) -- exact zeros, excluded from the clip maximum.
  const int clipN = per_clip ? nv : N;
  const int clip_frames = per_clip ? nv / HOP : n_frames;

  for (int i = tid; i < FR * NFFT; i += 256) {
    const int r = i / NFFT, n = i % NFFT;
    const int f = f0 + r;
    float v = 0.f;
    if (f < clip_frames) {
      int j = f * HOP + n - NFFT / 2;
      if (j < 0) j = -j;
      if (j >= clipN) j = 2 * (clipN - 1) - j;
      v = (j >= 0 && j < nv) ? x[j] * window[n] : 0.f;
    }
    lds[r * XLD + n] = v;
  }
  __syncthreads();

  // ---- DFT on the fp32 MFMA: each wave owns bin tiles bt = wave, wave+4, ... (13 tiles of 16 bins) ----
  f32x4_t pw[4][2];  // up to 4 bin tiles per wave x 2 frame fragments
  int nbt = 0;
  for (int bt = wave; bt < NBINP / 16; bt += 4, nbt++) {
    f32x4_t re[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    f32x4_t im[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    const float* tc = twiddle + bt * 16 + li;
    const float* ts = twiddle + NBINP + bt * 16 + li;
#pragma unroll 4
    for (int ks = 0; ks < NFFT / 4; ks++) {
      const int k = ks * 4 + g;
      const float a0 = lds[li * XLD + k];
      const float a1 = lds[(16 + li) * XLD + k];
      const float c = tc[k * TWC];
      const float s = ts[k * TWC];
      // operands swapped so that lane owns frame (l&15) x 4 consecutive bins
      re[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c, a0, re[0], 0, 0, 0);
      re[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c, a1, re[1], 0, 0, 0);
      im[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(s, a0, im[0], 0, 0, 0);
      im[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(s, a1, im[1], 0, 0, 0);
    }
#pragma unroll
    for (int af = 0; af < 2; af++)
#pragma unroll
      for (int r = 0; r < 4; r++) pw[nbt][af][r] = re[af][r] * re[af][r] + im[af][r] * im[af][r];
  }
  __syncthreads();  // everyone is done reading the frames
  {
    int i = 0;
    for (int bt = wave; bt < NBINP / 16; bt += 4, i++) {
#pragma unroll
      for (int af = 0; af < 2; af++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          // D[row = bin 4g+r][col = frame li]
          lds[(af * 16 + li) * PLD + bt * 16 + 4 * g + r] = pw[i][af][r];
        }
    }
  }
  __syncthreads();

  // ---- mel projection + log10 ----
  float lmax = -INFINITY;
  for (int idx = tid; idx < FR * n_mels; idx += 256) {
    const int r = idx / n_mels, m = idx % n_mels;
    const int f = f0 + r;
    if (f >= n_frames) continue;
    if (f >= clip_frames) {
      out[((int64_t)b * n_frames + f) * n_mels + m] = 0.f;
      continue;
    }
    float acc = 0.f;
    for (int k = 0; k < NBIN; k++) acc = fmaf(lds[r * PLD + k], melT[k * n_mels + m], acc);
    const float lv = log10f(fmaxf(acc, 1e-10f));
    out[((int64_t)b * n_frames + f) * n_mels + m] = lv;
    lmax = fmaxf(lmax, lv);
  }
  lmax = block_max<256>(lmax, red);
  if (tid == 0 && lmax > -INFINITY) atomicMax(clipmax + b, f2ord(lmax));
}

__global__ __launch_bounds__(256) void logmel_finish_kernel(float* __restrict__ out, int64_t elems_per_clip,
                                                            const int* __restrict__ clipmax,
                                                            const int* __restrict__ n_valid, int n_mels, int N,
                                                            int per_clip) {
  const int b = blockIdx.y;
  const float floorv = ord2f(clipmax[b]) - 8.0f;
  float* o = out + (int64_t)b * elems_per_clip;
  int64_t live = elems_per_clip;  // elements that belong to real frames of this clip
  if (per_clip) live = (int64_t)(min(n_valid[b], N) / HOP) * n_mels;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < live; i += (int64_t)gridDim.x * 256)
    o[i] = (fmaxf(o[i], floorv) + 4.0f) / 4.0f;
}

}  // namespace

extern "C" int slam_logmel_workspace_bytes(int64_t B) { return (int)(B * sizeof(int)); }

extern "C" int slam_logmel_fwd(const float* audio, int64_t ld_audio, const int32_t* n_valid,
                               int64_t n_samples, const float* window400, const float* twiddle_400x416,
                               const float* mel_filters_T, int64_t n_mels, float* out_mel,
                               int32_t* workspace, int64_t B, int per_clip, void* stream) {
  SLAM_CHECK_ARG(!per_clip || n_valid, "slam_logmel_fwd: per_clip mode needs n_valid");
  SLAM_CHECK_ARG(audio && window400 && twiddle_400x416 && mel_filters_T && out_mel && workspace,
                 "slam_logmel_fwd: null pointer");
  SLAM_CHECK_ARG(B > 0 && B < 65536, "slam_logmel_fwd: bad batch %ld", (long)B);
  SLAM_CHECK_ARG(n_samples >= 400 && n_samples % 160 == 0 && n_samples < (1ll << 30),
                 "slam_logmel_fwd: n_samples=%ld must be a multiple of 160 (hop) and >= 400", (long)n_samples);
  SLAM_CHECK_ARG(n_mels > 0 && n_mels <= 256, "slam_logmel_fwd: n_mels=%ld out of range", (long)n_mels);
  SLAM_CHECK_ARG(ld_audio >= 1, "slam_logmel_fwd: bad ld_audio");
  const int n_frames = (int)(n_samples / 160);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(logmel_init_kernel, dim3((unsigned)cdiv64(B, 256)), dim3(256), 0, s, workspace, (int)B);
  dim3 grid((unsigned)cdiv64(n_frames, FR), (unsigned)B);
  hipLaunchKernelGGL(logmel_stft_kernel, grid, dim3(256), 0, s, audio, ld_audio, n_valid, (int)n_samples,
                     window400, twiddle_400x416, mel_filters_T, (int)n_mels, out_mel, n_frames, workspace, per_clip);
  const int64_t per_clip_elems = (int64_t)n_frames * n_mels;
  dim3 grid2((unsigned)std::min<int64_t>(cdiv64(per_clip_elems, 256), 1024), (unsigned)B);
  hipLaunchKernelGGL(logmel_finish_kernel, grid2, dim3(256), 0, s, out_mel, per_clip_elems, workspace, n_valid, (int)n_mels,
                     (int)n_samples, per_clip);
  SLAM_CHECK_LAUNCH("slam_logmel_fwd");
  return 0;
}
