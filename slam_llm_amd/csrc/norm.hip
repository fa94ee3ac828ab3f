// HBM-bound row normalisations: LayerNorm (Whisper encoder) and RMSNorm fwd/bwd (Llama).
//
// Reference semantics:
//  * openai-whisper LayerNorm computes in fp32 and casts back (called from
//    src/slam_llm/models/encoder.py:26-29 via the blocks and ln_post).
//  * HF LlamaRMSNorm: variance in fp32, x*rsqrt(var+eps) cast back to the input dtype, THEN multiplied
//    by the weight (transformers/models/llama/modeling_llama.py:62-67) -> two roundings, reproduced here.
// One wave (64 lanes) per row, 16-byte (8 x bf16) accesses, 4 rows per 256-thread workgroup; rows are
// re-read from L1/L2 for the second/third pass, so HBM traffic is one read + one write per element.
#include "common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;
constexpr int LN_NARROW_MIN_ROWS = 4096;   // LayerNorm: batches of at least this many rows of d <= 1536 take the two-rows-per-wave kernel
template <bool CACHE> struct RowUnroll { static constexpr int N = CACHE ? 8 : 1; };

// CACHE: the row (d <= 4096 -> at most 8 16-byte chunks per lane) stays in registers between the passes instead of being
// re-read from L1/L2: one global load per element, half the load instructions (rmsnorm bwd 90 -> ~70 us on 11780 x 4096).
template <bool CACHE>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ b,
                                                        bf16_t* __restrict__ y, int64_t ldy, int M,
                                                        int d, float eps, int act, float* __restrict__ mean_out,
                                                        float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= M) return;
  const bf16_t* xr = x + (int64_t)row * ldx;
  const int nch = d >> 3;
  u16x8_t xc[CACHE ? 8 : 1];
  float s = 0.f;
#pragma unroll RowUnroll<CACHE>::N
  for (int i = 0; i < (CACHE ? 8 : (nch + 63) / 64); i++) {
    const int c = lane + i * 64;
    if (c >= nch) break;
    const u16x8_t v = *reinterpret_cast<const u16x8_t*>(xr + c * 8);
    if constexpr (CACHE) xc[i] = v;
#pragma unroll
    for (int e = 0; e < 8; e++) s += bf2f(v[e]);
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll RowUnroll<CACHE>::N
  for (int i = 0; i < (CACHE ? 8 : (nch + 63) / 64); i++) {
    const int c = lane + i * 64;
    if (c >= nch) break;
    const u16x8_t v = CACHE ? xc[CACHE ? i : 0] : *reinterpret_cast<const u16x8_t*>(xr + c * 8);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float t = bf2f(v[e]) - mean;
      q += t * t;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
  if (lane == 0 && mean_out) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
  bf16_t* yr = y + (int64_t)row * ldy;
#pragma unroll RowUnroll<CACHE>::N
  for (int i = 0; i < (CACHE ? 8 : (nch + 63) / 64); i++) {
    const int c = lane + i * 64;
    if (c >= nch) break;
    const u16x8_t v = CACHE ? xc[CACHE ? i : 0] : *reinterpret_cast<const u16x8_t*>(xr + c * 8);
    const float4 w0 = *reinterpret_cast<const float4*>(w + c * 8);
    const float4 w1 = *reinterpret_cast<const float4*>(w + c * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(b + c * 8);
    const float4 b1 = *reinterpret_cast<const float4*>(b + c * 8 + 4);
    const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    u16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      float t = (bf2f(v[e]) - mean) * rstd * ww[e] + bb[e];
      if (act == 1) t = 0.5f * t * (1.0f + erff(t * 0.70710678118654752440f));  // exact GELU (HuBERT conv layers)
      o[e] = f2bf(t);
    }
    *reinterpret_cast<u16x8_t*>(yr + c * 8) = o;
  }
}

// Narrow rows (d <= 1536: Whisper's 1280; at most NCH = 3 16-byte chunks per lane): RPW rows per wave, all their chunks requested before the first reduction.
// One wave per 2.5 KB row keeps too few bytes in flight per CU (4.7 TB/s on 46500 x 1280); per row the same lane -> chunk map and the same order of
// additions as layernorm_kernel: bit-identical results.
template <int NCH, int RPW>
__global__ __launch_bounds__(256) void layernorm_narrow_kernel(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ w, const float* __restrict__ b,
                                                               bf16_t* __restrict__ y, int64_t ldy, int M, int d, float eps, int act,
                                                               float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6)) * RPW;
  if (row0 >= M) return;
  const int nch = d >> 3;
  u16x8_t xc[RPW][NCH];
#pragma unroll
  for (int r = 0; r < RPW; r++) {
    const bf16_t* xr = x + (int64_t)min(row0 + r, M - 1) * ldx;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      const int c = lane + i * 64;
      if (c < nch) xc[r][i] = *reinterpret_cast<const u16x8_t*>(xr + c * 8);
    }
  }
  float mean[RPW], rstd[RPW];
#pragma unroll
  for (int r = 0; r < RPW; r++) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      if (lane + i * 64 < nch) {
#pragma unroll
        for (int e = 0; e < 8; e++) s += bf2f(xc[r][i][e]);
      }
    }
    mean[r] = wave_sum(s) / (float)d;
  }
#pragma unroll
  for (int r = 0; r < RPW; r++) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      if (lane + i * 64 < nch) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float t = bf2f(xc[r][i][e]) - mean[r];
          q += t * t;
        }
      }
    }
    rstd[r] = rsqrtf(wave_sum(q) / (float)d + eps);
  }
#pragma unroll
  for (int i = 0; i < NCH; i++) {
    const int c = lane + i * 64;
    if (c >= nch) continue;
    const float4 w0 = *reinterpret_cast<const float4*>(w + c * 8);
    const float4 w1 = *reinterpret_cast<const float4*>(w + c * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(b + c * 8);
    const float4 b1 = *reinterpret_cast<const float4*>(b + c * 8 + 4);
    const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int r = 0; r < RPW; r++) {
      if (row0 + r >= M) continue;
      u16x8_t o;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float t = (bf2f(xc[r][i][e]) - mean[r]) * rstd[r] * ww[e] + bb[e];
        if (act == 1) t = 0.5f * t * (1.0f + erff(t * 0.70710678118654752440f));
        o[e] = f2bf(t);
      }
      *reinterpret_cast<u16x8_t*>(y + (int64_t)(row0 + r) * ldy + c * 8) = o;
    }
  }
  if (lane == 0 && mean_out) {
#pragma unroll
    for (int r = 0; r < RPW; r++) {
      if (row0 + r < M) {
        mean_out[row0 + r] = mean[r];
        rstd_out[row0 + r] = rstd[r];
      }
    }
  }
}

template <bool CACHE>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ w,
                                                          bf16_t* __restrict__ y, int64_t ldy,
                                                          float* __restrict__ rstd_out, int M, int d,
                                                          float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= M) return;
  const bf16_t* xr = x + (int64_t)row * ldx;
  const int nch = d >> 3;
  u16x8_t xc[CACHE ? 8 : 1];
  float q = 0.f;
#pragma unroll RowUnroll<CACHE>::N
  for (int i = 0; i < (CACHE ? 8 : (nch + 63) / 64); i++) {
    const int c = lane + i * 64;
    if (c >= nch) break;
    const u16x8_t v = *reinterpret_cast<const u16x8_t*>(xr + c * 8);
    if constexpr (CACHE) xc[i] = v;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float t = bf2f(v[e]);
      q += t * t;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
  if (lane == 0 && rstd_out) rstd_out[row] = rstd;
  bf16_t* yr = y + (int64_t)row * ldy;
#pragma unroll RowUnroll<CACHE>::N
  for (int i = 0; i < (CACHE ? 8 : (nch + 63) / 64); i++) {
    const int c = lane + i * 64;
    if (c >= nch) break;
    const u16x8_t v = CACHE ? xc[CACHE ? i : 0] : *reinterpret_cast<const u16x8_t*>(xr + c * 8);
    const float4 w0 = *reinterpret_cast<const float4*>(w + c * 8);
    const float4 w1 = *reinterpret_cast<const float4*>(w + c * 8 + 4);
    const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    u16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float xn = bf2f(f2bf(bf2f(v[e]) * rstd));  // cast back to the input dtype first
      o[e] = f2bf(ww[e] * xn);
    }
    *reinterpret_cast<u16x8_t*>(yr + c * 8) = o;
  }
}

// few rows (decode: batch x beams): one workgroup per row so the row is read with one load per thread and kept in
// registers between the two passes (the one-wave-per-row kernel above serialises 8 dependent loads per pass here)
__global__ __launch_bounds__(256) void rmsnorm_fwd_small_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                                const float* __restrict__ w,
                                                                bf16_t* __restrict__ y, int64_t ldy,
                                                                float* __restrict__ rstd_out, int d, float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const bf16_t* xr = x + (int64_t)row * ldx;
  const int nch = d >> 3;
  constexpr int MAXC = 4;  // d <= 8192
  u16x8_t v[MAXC];
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; i++) {
    const int c = tid + i * 256;
    if (c < nch) {
      v[i] = *reinterpret_cast<const u16x8_t*>(xr + c * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float t = bf2f(v[i][e]);
        q += t * t;
      }
    }
  }
  const float rstd = rsqrtf(block_sum<256>(q, red) / (float)d + eps);
  if (tid == 0 && rstd_out) rstd_out[row] = rstd;
  bf16_t* yr = y + (int64_t)row * ldy;
#pragma unroll
  for (int i = 0; i < MAXC; i++) {
    const int c = tid + i * 256;
    if (c < nch) {
      const float4 w0 = *reinterpret_cast<const float4*>(w + c * 8);
      const float4 w1 = *reinterpret_cast<const float4*>(w + c * 8 + 4);
      const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      u16x8_t o;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float xn = bf2f(f2bf(bf2f(v[i][e]) * rstd));  // cast back to the input dtype first
        o[e] = f2bf(ww[e] * xn);
      }
      *reinterpret_cast<u16x8_t*>(yr + c * 8) = o;
    }
  }
}

// dx = rstd * (g - xhat * mean(g * xhat)) [* gscale] + dres ,  g = dy * w, xhat = x * rstd
template <bool CACHE>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(
    const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ rstd_in,
    const float* __restrict__ w, const bf16_t* __restrict__ dy, int64_t lddy,
    const bf16_t* __restrict__ dres, int64_t lddres, bf16_t* __restrict__ dx, int64_t lddx,
    const float* __restrict__ gscale, int M, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= M) return;
  const bf16_t* xr = x + (int64_t)row * ldx;
  const bf16_t* dyr = dy + (int64_t)row * lddy;
  const float rstd = rstd_in[row];
  const float gs = gscale ? *gscale : 1.0f;
  const int nch = d >> 3;
  float dot = 0.f;
  u16x8_t xc[CACHE ? 8 : 1], gc[CACHE ? 8 : 1];
#pragma unroll RowUnroll<CACHE>::N
  for (int i = 0; i < (CACHE ? 8 : (nch + 63) / 64); i++) {
    const int c = lane + i * 64;
    if (c >= nch) break;
    const u16x8_t v = *reinterpret_cast<const u16x8_t*>(xr + c * 8);
    const u16x8_t g = *reinterpret_cast<const u16x8_t*>(dyr + c * 8);
    if constexpr (CACHE) {
      xc[i] = v;
      gc[i] = g;
    }
    const float4 w0 = *reinterpret_cast<const float4*>(w + c * 8);
    const float4 w1 = *reinterpret_cast<const float4*>(w + c * 8 + 4);
    const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
    for (int e = 0; e < 8; e++) dot += bf2f(g[e]) * ww[e] * bf2f(v[e]) * rstd;
  }
  const float mdot = wave_sum(dot) / (float)d;
  bf16_t* dxr = dx + (int64_t)row * lddx;
#pragma unroll RowUnroll<CACHE>::N
  for (int i = 0; i < (CACHE ? 8 : (nch + 63) / 64); i++) {
    const int c = lane + i * 64;
    if (c >= nch) break;
    const u16x8_t v = CACHE ? xc[CACHE ? i : 0] : *reinterpret_cast<const u16x8_t*>(xr + c * 8);
    const u16x8_t g = CACHE ? gc[CACHE ? i : 0] : *reinterpret_cast<const u16x8_t*>(dyr + c * 8);
    const float4 w0 = *reinterpret_cast<const float4*>(w + c * 8);
    const float4 w1 = *reinterpret_cast<const float4*>(w + c * 8 + 4);
    const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    u16x8_t r;
    if (dres) r = *reinterpret_cast<const u16x8_t*>(dres + (int64_t)row * lddres + c * 8);
    u16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float xh = bf2f(v[e]) * rstd;
      float t = rstd * (bf2f(g[e]) * ww[e] - xh * mdot) * gs;
      if (dres) t += bf2f(r[e]);
      o[e] = f2bf(t);
    }
    *reinterpret_cast<u16x8_t*>(dxr + c * 8) = o;
  }
}

// LayerNorm backward (Q-Former projector, src/slam_llm/models/projector.py:51-80 -> HF Blip2QFormer LayerNorms):
//   g = dy * w ; dx = rstd * (g - mean(g) - xhat * mean(g * xhat))        (one wave per row)
__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                               const float* __restrict__ mean_in,
                                                               const float* __restrict__ rstd_in,
                                                               const float* __restrict__ w,
                                                               const bf16_t* __restrict__ dy, int64_t lddy,
                                                               bf16_t* __restrict__ dx, int64_t lddx, int M, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= M) return;
  const bf16_t* xr = x + (int64_t)row * ldx;
  const bf16_t* dyr = dy + (int64_t)row * lddy;
  const float mean = mean_in[row], rstd = rstd_in[row];
  const int nch = d >> 3;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < nch; c += 64) {
    const u16x8_t v = *reinterpret_cast<const u16x8_t*>(xr + c * 8);
    const u16x8_t g = *reinterpret_cast<const u16x8_t*>(dyr + c * 8);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float gw = bf2f(g[e]) * w[c * 8 + e];
      s1 += gw;
      s2 += gw * (bf2f(v[e]) - mean) * rstd;
    }
  }
  s1 = wave_sum(s1) / (float)d;
  s2 = wave_sum(s2) / (float)d;
  bf16_t* dxr = dx + (int64_t)row * lddx;
  for (int c = lane; c < nch; c += 64) {
    const u16x8_t v = *reinterpret_cast<const u16x8_t*>(xr + c * 8);
    const u16x8_t g = *reinterpret_cast<const u16x8_t*>(dyr + c * 8);
    u16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float xh = (bf2f(v[e]) - mean) * rstd;
      o[e] = f2bf(rstd * (bf2f(g[e]) * w[c * 8 + e] - s1 - xh * s2));
    }
    *reinterpret_cast<u16x8_t*>(dxr + c * 8) = o;
  }
}

// dgamma[c] (+)= sum_r dy[r,c] * xhat[r,c] ; dbeta[c] (+)= sum_r dy[r,c]   (64 columns per workgroup, fixed order)
__global__ __launch_bounds__(256) void layernorm_bwd_params_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                                   const float* __restrict__ mean_in,
                                                                   const float* __restrict__ rstd_in,
                                                                   const bf16_t* __restrict__ dy, int64_t lddy,
                                                                   float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, int64_t M, int d,
                                                                   int accumulate) {
  __shared__ float rg[32][65], rb[32][65];
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + tx * 8;
  float ag[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ab[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < d) {
    for (int64_t m = ty; m < M; m += 32) {
      const u16x8_t v = *reinterpret_cast<const u16x8_t*>(x + m * ldx + c0);
      const u16x8_t g = *reinterpret_cast<const u16x8_t*>(dy + m * lddy + c0);
      const float mean = mean_in[m], rstd = rstd_in[m];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float gf = bf2f(g[e]);
        ag[e] += gf * (bf2f(v[e]) - mean) * rstd;
        ab[e] += gf;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; e++) {
    rg[ty][tx * 8 + e] = ag[e];
    rb[ty][tx * 8 + e] = ab[e];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < d) {
      float sg = 0.f, sb = 0.f;
#pragma unroll
      for (int r = 0; r < 32; r++) {
        sg += rg[r][threadIdx.x];
        sb += rb[r][threadIdx.x];
      }
      dgamma[c] = accumulate ? dgamma[c] + sg : sg;
      dbeta[c] = accumulate ? dbeta[c] + sb : sb;
    }
  }
}

}  // namespace

extern "C" int slam_layernorm_fwd(const void* x, int64_t ldx, const float* weight, const float* bias,
                                  void* y, int64_t ldy, int64_t M, int64_t d, float eps, int act, float* mean_out,
                                  float* rstd_out, void* stream) {
  SLAM_CHECK_ARG((mean_out == nullptr) == (rstd_out == nullptr), "slam_layernorm_fwd: mean/rstd outputs must both be set or both null");
  SLAM_CHECK_ARG(act == 0 || act == 1, "slam_layernorm_fwd: act %d unknown (0 none, 1 gelu)", act);
  SLAM_CHECK_ARG(x && weight && bias && y, "slam_layernorm_fwd: null pointer");
  SLAM_CHECK_ARG(M > 0 && d > 0 && d % 8 == 0, "slam_layernorm_fwd: bad shape M=%ld d=%ld (d%%8 must be 0)", (long)M, (long)d);
  SLAM_CHECK_ARG(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= d && ldy >= d, "slam_layernorm_fwd: bad leading dims");
  const unsigned grid = (unsigned)cdiv64(M, ROWS_PER_BLOCK);
  // narrow rows of a large batch: two rows per wave (round 6: 46500 x 1280 in the C3 step 50.2 -> 44.6 us per launch under rocprofv3, 4.74 -> 5.34 TB/s; four rows per
  // wave: no better than one; bit-identical to the one-row kernel, tests/test_ops_gpu.py::test_layernorm_two_rows_per_wave_is_bit_identical)
  if (d <= 1536 && M >= LN_NARROW_MIN_ROWS)
    hipLaunchKernelGGL((layernorm_narrow_kernel<3, 2>), dim3((unsigned)cdiv64(M, ROWS_PER_BLOCK * 2)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       ldx, weight, bias, (bf16_t*)y, ldy, (int)M, (int)d, eps, act, mean_out, rstd_out);
  else if (d <= 4096)
    hipLaunchKernelGGL(layernorm_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       ldx, weight, bias, (bf16_t*)y, ldy, (int)M, (int)d, eps, act, mean_out, rstd_out);
  else
    hipLaunchKernelGGL(layernorm_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       ldx, weight, bias, (bf16_t*)y, ldy, (int)M, (int)d, eps, act, mean_out, rstd_out);
  SLAM_CHECK_LAUNCH("slam_layernorm_fwd");
  return 0;
}

extern "C" int slam_rmsnorm_fwd(const void* x, int64_t ldx, const float* weight, void* y, int64_t ldy,
                                float* rstd, int64_t M, int64_t d, float eps, void* stream) {
  SLAM_CHECK_ARG(x && weight && y, "slam_rmsnorm_fwd: null pointer");
  SLAM_CHECK_ARG(M > 0 && d > 0 && d % 8 == 0, "slam_rmsnorm_fwd: bad shape M=%ld d=%ld (d%%8 must be 0)", (long)M, (long)d);
  SLAM_CHECK_ARG(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= d && ldy >= d, "slam_rmsnorm_fwd: bad leading dims");
  const unsigned grid = (unsigned)cdiv64(M, ROWS_PER_BLOCK);
  if (M <= 128 && d <= 8192)
    hipLaunchKernelGGL(rmsnorm_fwd_small_kernel, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, ldx, weight, (bf16_t*)y, ldy, rstd, (int)d, eps);
  else
    hipLaunchKernelGGL(d <= 4096 ? rmsnorm_fwd_kernel<true> : rmsnorm_fwd_kernel<false>, dim3(grid), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)x, ldx, weight, (bf16_t*)y, ldy, rstd, (int)M, (int)d, eps);
  SLAM_CHECK_LAUNCH("slam_rmsnorm_fwd");
  return 0;
}

extern "C" int slam_rmsnorm_bwd(const void* x, int64_t ldx, const float* rstd, const float* weight,
                                const void* dy, int64_t lddy, const void* dres, int64_t lddres,
                                void* dx, int64_t lddx, const float* grad_scale, int64_t M, int64_t d,
                                void* stream) {
  SLAM_CHECK_ARG(x && rstd && weight && dy && dx, "slam_rmsnorm_bwd: null pointer");
  SLAM_CHECK_ARG(M > 0 && d > 0 && d % 8 == 0, "slam_rmsnorm_bwd: bad shape M=%ld d=%ld", (long)M, (long)d);
  SLAM_CHECK_ARG(ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && (!dres || lddres % 8 == 0),
                 "slam_rmsnorm_bwd: leading dims must be multiples of 8");
  const unsigned grid = (unsigned)cdiv64(M, ROWS_PER_BLOCK);
  hipLaunchKernelGGL(d <= 4096 ? rmsnorm_bwd_kernel<true> : rmsnorm_bwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ldx, rstd, weight, (const bf16_t*)dy, lddy, (const bf16_t*)dres,
                     lddres, (bf16_t*)dx, lddx, grad_scale, (int)M, (int)d);
  SLAM_CHECK_LAUNCH("slam_rmsnorm_bwd");
  return 0;
}

extern "C" int slam_layernorm_bwd(const void* x, int64_t ldx, const float* mean, const float* rstd, const float* weight,
                                  const void* dy, int64_t lddy, void* dx, int64_t lddx, float* dgamma, float* dbeta,
                                  int64_t M, int64_t d, int accumulate, void* stream) {
  SLAM_CHECK_ARG(x && mean && rstd && weight && dy, "slam_layernorm_bwd: null pointer");
  SLAM_CHECK_ARG(M > 0 && d > 0 && d % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0 && (!dx || lddx % 8 == 0), "slam_layernorm_bwd: bad shape");
  SLAM_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "slam_layernorm_bwd: dgamma/dbeta must both be set or both null");
  hipStream_t s = (hipStream_t)stream;
  if (dx)
    hipLaunchKernelGGL(layernorm_bwd_dx_kernel, dim3((unsigned)cdiv64(M, ROWS_PER_BLOCK)), dim3(256), 0, s, (const bf16_t*)x, ldx,
                       mean, rstd, weight, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, (int)M, (int)d);
  if (dgamma)
    hipLaunchKernelGGL(layernorm_bwd_params_kernel, dim3((unsigned)cdiv64(d, 64)), dim3(256), 0, s, (const bf16_t*)x, ldx, mean,
                       rstd, (const bf16_t*)dy, lddy, dgamma, dbeta, M, (int)d, accumulate);
  SLAM_CHECK_LAUNCH("slam_layernorm_bwd");
  return 0;
}
