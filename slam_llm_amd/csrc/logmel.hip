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
// Kernel 1 (round 4): the 400-point real DFT as a FOLDED product on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32: bitwise an fmaf chain).
// cos(2 pi k (400 - n) / 400) = cos(2 pi k n / 400) and sin(...) = -sin(...), so with e[n] = xw[n] + xw[400 - n] (n = 1..199; e[0] =
// xw[0], e[200] = xw[200]) and o[n] = xw[n] - xw[400 - n]:   Re X[k] = sum_{n=0..200} e[n] cos(2 pi k n / 400),   Im X[k] = -sum_{n=1..199}
// o[n] sin(2 pi k n / 400)  -- half the multiplies of the plain [32 x 400] . [400 x 416] product this replaces (0.11 TB/s, ~1 ms for 31
// clips; its twiddles were fetched as two 4-byte loads per MFMA pair).  One workgroup (8 waves) walks 32-frame blocks of one clip:
//   * the block's 5360-sample span is staged raw in LDS (requested one block ahead), reflected / clipped exactly like the plain form;
//   * the waves fold it into LDS rows [e | o] (208 + 208 k-values per frame, k-value n = 4 ks + g stored at g * 52 + ks so that a lane
//     reads four k-steps as one 16-byte LDS read; row pitch 420 floats = conflict-free);
//   * wave w owns bin tiles {w, w + 8} of the 13 (16 bins each); the twiddles stream from an L2-resident host table laid out
//     [tile][cos | sin][k-quad][lane][4] (one 16-byte load per lane and k-quad, 346 KB);
//   * |X|^2 goes back to LDS over the frame rows, the mel projection runs over each filter's non-zero bin range only (bit-identical to
//     the dense dot: the skipped terms are fmaf(p, 0, acc)), log10, store, ordered-int atomicMax of the clip maximum.
// Kernel 2 applies the floor and the affine map in place.
// ------------------------------------------------------------------------------------------

#include "common.h"
#include <limits.h>
#include <algorithm>
#include <type_traits>

namespace {

constexpr int NFFT = 400, HOP = 160, NBIN = 201, NTILE = 13, KQ = 13, FR = 32, ROW = 420, PLD = 209, SPAN = (FR - 1) * HOP + NFFT;
constexpr int LM_THREADS = 512, MELV = 512;

__device__ __forceinline__ int f2ord(float f) {
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// workspace: [B] clip maxima (ordered ints), then -- 16-byte aligned -- the mel filterbank in sparse form, built once per launch by
// workgroup 0 of the init kernel and copied into LDS by every STFT workgroup: MelPack
struct MelPack {
  unsigned short lo[256], n[256], off[256];        // filter m: bins lo .. lo + n, values at val[off ..]
  int packed, pad[3];                              // 0: the ranges do not fit MELV values -> the dense table is read instead
  float val[MELV];
};
__host__ __device__ inline int64_t melpack_offset(int64_t B) { return (B * 4 + 15) / 16 * 16; }

__global__ __launch_bounds__(1024) void logmel_init_kernel(int* clipmax, int B, const float* __restrict__ melT, int n_mels, MelPack* mp) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) clipmax[i] = INT_MIN;
  if (blockIdx.x != 0) return;
  // the non-zero bin range of every filter (the slaney triangles touch 2 .. ~25 of the 201 bins): thread (m, part) takes every
  // parts-th bin, the ranges meet through LDS atomics
  __shared__ int mlo[256], mhi[256];
  const int tid = threadIdx.x;
  if (tid < 256) { mlo[tid] = NBIN; mhi[tid] = 0; }
  __syncthreads();
  const int parts = 1024 / n_mels, m = tid % n_mels, part = tid / n_mels;
  if (part < parts) {
    int lo = NBIN, hi = 0;
#pragma unroll 8
    for (int k = part; k < NBIN; k += parts) {
      const bool nz = melT[k * n_mels + m] != 0.f;
      lo = min(lo, nz ? k : NBIN);
      hi = max(hi, nz ? k + 1 : 0);
    }
    if (hi > 0) { atomicMin(mlo + m, lo); atomicMax(mhi + m, hi); }
  }
  __syncthreads();
  if (tid < n_mels) {
    int off = 0;
    for (int j = 0; j < tid; j++) off += mhi[j] - min(mlo[j], mhi[j]);
    const int lo = min(mlo[tid], mhi[tid]), n = mhi[tid] - lo;
    mp->lo[tid] = (unsigned short)lo;
    mp->n[tid] = (unsigned short)n;
    mp->off[tid] = (unsigned short)min(off, 65535);
    if (tid == n_mels - 1) mp->packed = (off + n <= MELV) ? 1 : 0;
    if (off + n <= MELV)
      for (int i = 0; i < n; i++) mp->val[off + i] = melT[(lo + i) * n_mels + tid];
  }
}

__global__ __launch_bounds__(LM_THREADS, 4) void logmel_stft_kernel(const float* __restrict__ audio, int64_t ld_audio,
                                                                     const int* __restrict__ n_valid, int N,
                                                                     const float* __restrict__ window,
                                                                     const float* __restrict__ twiddle,
                                                                     const float* __restrict__ melT, int n_mels,
                                                                     float* __restrict__ out, int n_frames,
                                                                     int* __restrict__ clipmax, const MelPack* __restrict__ mp,
                                                                     int per_clip) {
  __shared__ __attribute__((aligned(16))) float frames[FR * ROW];   // [32][e 208 | o 208 (+4)]; later the power rows [32][209]
  __shared__ float raw[SPAN];
  __shared__ unsigned short mel_lo[256], mel_n[256], mel_off[256];
  __shared__ float mel_val[MELV];                  // the filters' non-zero ranges, packed (slaney: ~2 x 201 values)
  __shared__ float red[LM_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int b = blockIdx.y;
  const float* x = audio + (int64_t)b * ld_audio;
  const int nv = n_valid ? min(n_valid[b], N) : N;
  // per_clip: the clip is NOT padded to N first (pad_or_trim off, speech_dataset_large.py:102-104): its own length is
  // the STFT length (reflection about its own end), it owns nv/160 frames and the rest of the row is mel-space zero
  // padding added by the collator (speech_dataset_large.py:194-197) -- exact zeros, excluded from the clip maximum.
  const int clipN = per_clip ? nv : N;
  const int clip_frames = per_clip ? nv / HOP : n_frames;
  const int nblk = (n_frames + FR - 1) / FR;

  // the filterbank in sparse form (built by the init kernel): one coalesced copy into LDS
  for (int i = tid; i < 256; i += LM_THREADS) { mel_lo[i] = mp->lo[i]; mel_n[i] = mp->n[i]; mel_off[i] = mp->off[i]; }
  for (int i = tid; i < MELV; i += LM_THREADS) mel_val[i] = mp->val[i];
  const bool packed = mp->packed != 0;
  __syncthreads();

  // sample i of block blk's span = position blk * 32 * 160 - 200 + i of the (reflect-padded, zero-extended) clip
  constexpr int RPT = (SPAN + LM_THREADS - 1) / LM_THREADS;     // 11 staged samples per thread
  float stage[RPT];
  auto request = [&](int blk) {
#pragma unroll
    for (int q = 0; q < RPT; q++) {
      const int i = q * LM_THREADS + tid;
      int j = blk * FR * HOP - NFFT / 2 + i;
      if (j < 0) j = -j;
      if (j >= clipN) j = 2 * (clipN - 1) - j;
      stage[q] = (i < SPAN && j >= 0 && j < nv) ? x[j] : 0.f;
    }
  };
  // fold: thread = one k-value n (two window taps in registers for the whole launch) x every second frame
  const int fn = tid % 208, fr0 = tid / 208;       // threads 416 .. 511 idle in this phase
  const float w_a = (fr0 < 2 && fn <= 200) ? window[fn] : 0.f;
  const float w_c = (fr0 < 2 && fn >= 1 && fn < 200) ? window[NFFT - fn] : 0.f;
  const int fslot = (fn & 3) * 52 + (fn >> 2);
  // products: 26 tasks (bin tile, cos | sin): waves' 0 and 1 carry four, the other six three; in odd workgroups the four-task waves
  // sit on the other SIMD pair (two workgroups share a CU)
  const int wv = (wave + (blockIdx.x & 1) * 2) & 7;
  // mel: thread = one filter (its range in registers) x every groups-th frame
  // (filters dealt round-robin to the waves of a group: the wide high filters do not all land in one wave)
  const int mgroups = LM_THREADS / n_mels;
  const int mt = tid % n_mels, mr0 = tid / n_mels;
  const int mwv = (n_mels % 64 == 0) ? n_mels / 64 : 1;
  const int mm = (mwv > 1) ? (mt & 63) * mwv + (mt >> 6) : mt;
  const int m_lo = mel_lo[mm], m_cnt = (mr0 < mgroups) ? mel_n[mm] : 0;
  const float* m_val = mel_val + mel_off[mm];
  float lmax = -INFINITY;

  int blk = blockIdx.x;
  if (blk < nblk) request(blk);
  for (; blk < nblk; blk += gridDim.x) {
    const int f0 = blk * FR;
    __syncthreads();                                   // the previous block's mel phase has read the power rows / raw is free
#pragma unroll
    for (int q = 0; q < RPT; q++) {
      const int i = q * LM_THREADS + tid;
      if (i < SPAN) raw[i] = stage[q];
    }
    __syncthreads();
    // ---- fold: frames[r][part * 208 + (n & 3) * 52 + (n >> 2)] ----
    if (fr0 < 2) {
#pragma unroll 4
      for (int r = fr0; r < FR; r += 2) {
        float e = 0.f, o = 0.f;
        if (f0 + r < clip_frames && fn <= 200) {
          const float a = raw[r * HOP + fn] * w_a;
          const float c = (fn >= 1 && fn < 200) ? raw[r * HOP + NFFT - fn] * w_c : 0.f;
          e = a + c;                                   // n = 0 and n = 200 have no partner (c = 0)
          o = (fn >= 1 && fn < 200) ? a - c : 0.f;
        }
        float* row = frames + r * ROW + fslot;
        row[0] = e;
        row[208] = o;
      }
    }
    __syncthreads();
    // ---- the folded products on the fp32 MFMA: D[row = bin 4 g + r][col = frame li] ----
    // wave' = 2 base + part: the cos (part 0) or sin (part 1) product of bin tiles base, base + 4, base + 8 (and 12 for base 0), two
    // tiles at a time against the same frame operands = four independent accumulation chains
    f32x4_t acc[4][2];                               // [tile slot][frame fragment]
#pragma unroll
    for (int ti = 0; ti < 4; ti++) acc[ti][0] = acc[ti][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int part = wv & 1, base = wv >> 1;
    const float* arow = frames + li * ROW + part * 208 + g * 52;
#pragma unroll
    for (int pr = 0; pr < 2; pr++) {
      const int tA = base + 8 * pr, tB = base + 8 * pr + 4;
      const bool two = tB < NTILE;                   // wave-uniform (tile 12 exists for base 0 only; 16 never)
      if (tA >= NTILE) continue;
      const float4* twA = reinterpret_cast<const float4*>(twiddle) + (int64_t)(2 * tA + part) * KQ * 64 + lane;
      const float4* twB = reinterpret_cast<const float4*>(twiddle) + (int64_t)(2 * (two ? tB : tA) + part) * KQ * 64 + lane;
      float4 nA = twA[0], nB = twB[0];               // twiddles one k-quad ahead
#pragma unroll 1
      for (int kq = 0; kq < KQ; kq++) {
        const float4 bA = nA, bB = nB;
        const int nx = min(kq + 1, KQ - 1) * 64;
        nA = twA[nx];
        nB = twB[nx];
        const float4 a0 = *reinterpret_cast<const float4*>(arow + 4 * kq);
        const float4 a1 = *reinterpret_cast<const float4*>(arow + 16 * ROW + 4 * kq);
        const float ba[4] = {bA.x, bA.y, bA.z, bA.w}, bb[4] = {bB.x, bB.y, bB.z, bB.w};
        const float aa0[4] = {a0.x, a0.y, a0.z, a0.w}, aa1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          acc[2 * pr][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ba[j], aa0[j], acc[2 * pr][0], 0, 0, 0);
          acc[2 * pr][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ba[j], aa1[j], acc[2 * pr][1], 0, 0, 0);
          if (two) {
            acc[2 * pr + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[j], aa0[j], acc[2 * pr + 1][0], 0, 0, 0);
            acc[2 * pr + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[j], aa1[j], acc[2 * pr + 1][1], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();                                   // everyone is done reading the frame rows
    if (blk + (int)gridDim.x < nblk) request(blk + gridDim.x);      // in flight under the power / mel phases (not live across the products)
    // |X|^2 over the frame rows in two steps: the cos waves store Re^2, then the sin waves add Im^2 (each (tile, part) has one owner)
#pragma unroll
    for (int step = 0; step < 2; step++) {
      if (part == step) {
#pragma unroll
        for (int ti = 0; ti < 4; ti++) {
          const int t = base + 4 * ((ti & 1) + 2 * (ti >> 1));     // slots: (pair 0: base, base + 4), (pair 1: base + 8, base + 12)
          if (t >= NTILE) continue;
#pragma unroll
          for (int f = 0; f < 2; f++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              float* pw = frames + (f * 16 + li) * PLD + t * 16 + 4 * g + r;
              const float sq = acc[ti][f][r] * acc[ti][f][r];
              *pw = step ? *pw + sq : sq;
            }
        }
      }
      __syncthreads();
    }

    // ---- mel projection over each filter's own bins + log10 ----
    // thread = filter mm x frames mr0, mr0 + mgroups, ...: the filter's terms outermost, so that the (up to 16) frames' chains run side
    // by side -- per frame the same fmaf order as a k-ascending dense dot, hence the same bits
    if (mr0 < mgroups) {
      const int nrr = (FR - mr0 + mgroups - 1) / mgroups;      // frames of this thread: 8 at 128 mels, 5 or 6 at 80
      const float* prow = frames + mr0 * PLD + m_lo;
      const int rstep = mgroups * PLD;
      auto project = [&](auto rr_c, int rr0) {
        constexpr int RR = decltype(rr_c)::value;
        // every read unconditional (rows past the thread's last frame are clamped onto it and their sums dropped): guarded reads compile
        // to one exec-masked block each, i.e. one exposed LDS round trip per term and frame (measured: 53 us of the 356)
        float am[RR];
#pragma unroll
        for (int rr = 0; rr < RR; rr++) am[rr] = 0.f;
        const float* pr[RR];
#pragma unroll
        for (int rr = 0; rr < RR; rr++) pr[rr] = prow + min(rr0 + rr, nrr - 1) * rstep;
        if (packed) {
          for (int i = 0; i < m_cnt; i++) {
            const float wgt = m_val[i];
#pragma unroll
            for (int rr = 0; rr < RR; rr++) am[rr] = fmaf(pr[rr][i], wgt, am[rr]);
          }
        } else {
          for (int i = 0; i < m_cnt; i++) {
            const float wgt = melT[(m_lo + i) * n_mels + mm];
#pragma unroll
            for (int rr = 0; rr < RR; rr++) am[rr] = fmaf(pr[rr][i], wgt, am[rr]);
          }
        }
#pragma unroll
        for (int rr = 0; rr < RR; rr++) {
          const int f = f0 + mr0 + (rr0 + rr) * mgroups;
          if (rr0 + rr >= nrr || f >= n_frames) continue;
          float* op = out + ((int64_t)b * n_frames + f) * n_mels + mm;
          if (f >= clip_frames) {
            *op = 0.f;
            continue;
          }
          // log10 through the hardware log2 (1 ulp): the fixture tolerance is 1e-4 on (log10 + 4) / 4
          const float lv = __log2f(fmaxf(am[rr], 1e-10f)) * 0.30102999566398120f;
          *op = lv;
          lmax = fmaxf(lmax, lv);
        }
      };
      project(std::integral_constant<int, 8>{}, 0);
      if (nrr > 8) project(std::integral_constant<int, 8>{}, 8);      // more than 128 filters: 11 or 16 frames per thread
    }
  }
  lmax = block_max<LM_THREADS>(lmax, red);
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

extern "C" int64_t slam_logmel_workspace_bytes(int64_t B) { return melpack_offset(B) + (int64_t)sizeof(MelPack); }

extern "C" int slam_logmel_fwd(const float* audio, int64_t ld_audio, const int32_t* n_valid,
                               int64_t n_samples, const float* window400, const float* twiddle_folded,
                               const float* mel_filters_T, int64_t n_mels, float* out_mel,
                               int32_t* workspace, int64_t B, int per_clip, void* stream) {
  SLAM_CHECK_ARG(!per_clip || n_valid, "slam_logmel_fwd: per_clip mode needs n_valid");
  SLAM_CHECK_ARG(audio && window400 && twiddle_folded && mel_filters_T && out_mel && workspace,
                 "slam_logmel_fwd: null pointer");
  SLAM_CHECK_ARG(B > 0 && B < 65536, "slam_logmel_fwd: bad batch %ld", (long)B);
  SLAM_CHECK_ARG(n_samples >= 400 && n_samples % 160 == 0 && n_samples < (1ll << 30),
                 "slam_logmel_fwd: n_samples=%ld must be a multiple of 160 (hop) and >= 400", (long)n_samples);
  SLAM_CHECK_ARG(n_mels > 0 && n_mels <= 256, "slam_logmel_fwd: n_mels=%ld out of range", (long)n_mels);
  SLAM_CHECK_ARG(((uintptr_t)workspace % 16) == 0, "slam_logmel_fwd: workspace must be 16-byte aligned");
  SLAM_CHECK_ARG(ld_audio >= 1, "slam_logmel_fwd: bad ld_audio");
  const int n_frames = (int)(n_samples / 160);
  hipStream_t s = (hipStream_t)stream;
  MelPack* mp = reinterpret_cast<MelPack*>(reinterpret_cast<char*>(workspace) + melpack_offset(B));
  hipLaunchKernelGGL(logmel_init_kernel, dim3((unsigned)cdiv64(B, 1024)), dim3(1024), 0, s, workspace, (int)B, mel_filters_T, (int)n_mels, mp);
  // two workgroups per CU and NOT ONE MORE (a 527-workgroup grid on 512 slots ran a second, nearly empty round: 333 us against 2xx);
  // each workgroup walks blocks blockIdx.x, + gridDim.x, ... of its clip
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
               ? prop.multiProcessorCount : 256;
  }
  const int64_t nblk = cdiv64(n_frames, FR);
  dim3 grid((unsigned)std::min<int64_t>(nblk, std::max<int64_t>(1, (2 * n_cu) / B)), (unsigned)B);
  hipLaunchKernelGGL(logmel_stft_kernel, grid, dim3(LM_THREADS), 0, s, audio, ld_audio, n_valid, (int)n_samples,
                     window400, twiddle_folded, mel_filters_T, (int)n_mels, out_mel, n_frames, workspace, mp, per_clip);
  const int64_t per_clip_elems = (int64_t)n_frames * n_mels;
  dim3 grid2((unsigned)std::min<int64_t>(cdiv64(per_clip_elems, 256), 1024), (unsigned)B);
  hipLaunchKernelGGL(logmel_finish_kernel, grid2, dim3(256), 0, s, out_mel, per_clip_elems, workspace, n_valid, (int)n_mels,
                     (int)n_samples, per_clip);
  SLAM_CHECK_LAUNCH("slam_logmel_fwd");
  return 0;
}
