// HBM-bound data-movement / elementwise kernels of the SLAM hot path (gfx950).
//   * head transpose (+ RoPE) : Llama rotary embedding (HF apply_rotary_pos_emb / rotate_half,
//     transformers/models/llama/modeling_llama.py:130-160, positions = arange(T) per row as the
//     reference never passes position_ids, src/slam_llm/models/slam_model.py:400) fused with the
//     [B,T,H,D] -> [B,H,D,Tp] transposes the MFMA attention kernels consume.
//   * SwiGLU fwd/bwd          : LlamaMLP act_fn(gate)*up (modeling_llama.py:163-177).
//   * conv im2col             : Whisper conv1 (k3,p1) / conv2 (k3,s2,p1) -> implicit-GEMM operand
//     (src/slam_llm/models/encoder.py:18-19).
//   * embed + audio splice    : src/slam_llm/models/slam_model.py:370-392 without the .tolist() host
//     sync or the per-sample python loop; plus its backward (a row gather into the projector grad).
//   * generic bf16 transpose, fp32->bf16 cast.
// All kernels move 16 bytes per lane per access and never synchronise with the host.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------
// head transpose with optional in-place rotary embedding
// src: rows (b*T + t), columns col0 + h*D + d.  dstT: [B, H, D, Tp] (t contiguous), zero padded.
// ------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void head_rope_transpose_kernel(
    bf16_t* __restrict__ src, int64_t ld, int col0, const float* __restrict__ cosT,
    const float* __restrict__ sinT, float sin_sign, bf16_t* __restrict__ dstT, int T, int Tp, int H,
    const int* __restrict__ positions) {
  // LDS tile [64 t][D] without padding; the 16-byte chunk index is XOR-swizzled with 2*(t>>3) so that the transposing
  // 2-byte reads below (8 lanes = 8 different t-groups, same d) land in 8 disjoint bank groups instead of 2
  // (a padded row stride cannot do that: 8 rows of any 16-byte-aligned stride are 0 or 32 banks apart)
  constexpr int LDT = D;
  constexpr int CMASK = D / 8 - 1;
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * LDT];
  const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x;
  constexpr int HC = D / 16;  // 16-byte chunks in half a head
  for (int item = tid; item < 64 * HC; item += 256) {
    const int tt = item / HC, c = item % HC;
    const int t = t0 + tt;
    u16x8_t y1, y2;
    if (t < T) {
      bf16_t* p = src + (int64_t)(b * (int64_t)T + t) * ld + col0 + h * D;
      const u16x8_t x1 = *reinterpret_cast<const u16x8_t*>(p + c * 8);
      const u16x8_t x2 = *reinterpret_cast<const u16x8_t*>(p + D / 2 + c * 8);
      if (cosT) {
        // training: position = t (the reference never passes position_ids); generate(): HF derives
        // position_ids from the attention mask, the caller passes them explicitly
        const int pp = positions ? positions[(int64_t)b * T + t] : t;
        const float* cp = cosT + (int64_t)pp * (D / 2) + c * 8;
        const float* sp = sinT + (int64_t)pp * (D / 2) + c * 8;
        // 4 x 16-byte table loads (rows are 256-byte aligned): scalar cp[e] / sp[e] reads made this kernel load-issue bound
        const float4 c0 = *reinterpret_cast<const float4*>(cp), c1 = *reinterpret_cast<const float4*>(cp + 4);
        const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
        const float cs8[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float sn8[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        float r1[8], r2[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float cs = cs8[e], sn = sn8[e] * sin_sign;
          const float a = bf2f(x1[e]), bb = bf2f(x2[e]);
          r1[e] = a * cs - bb * sn;
          r2[e] = bb * cs + a * sn;
        }
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        u32x4_t w1, w2;   // hardware round-to-nearest-even packs (v_cvt_pk_bf16_f32)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          w1[e] = pack2bf(r1[2 * e], r1[2 * e + 1]);
          w2[e] = pack2bf(r2[2 * e], r2[2 * e + 1]);
        }
        y1 = __builtin_bit_cast(u16x8_t, w1);
        y2 = __builtin_bit_cast(u16x8_t, w2);
        *reinterpret_cast<u16x8_t*>(p + c * 8) = y1;
        *reinterpret_cast<u16x8_t*>(p + D / 2 + c * 8) = y2;
      } else {
        y1 = x1;
        y2 = x2;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++) { y1[e] = 0; y2[e] = 0; }
    }
    if (dstT) {
      const int sw = ((tt >> 3) * 2) & CMASK;
      *reinterpret_cast<u16x8_t*>(&tile[tt * LDT + ((c ^ sw) << 3)]) = y1;
      *reinterpret_cast<u16x8_t*>(&tile[tt * LDT + (((HC + c) ^ sw) << 3)]) = y2;
    }
  }
  if (!dstT) return;
  __syncthreads();
  bf16_t* out = dstT + ((int64_t)(b * (int64_t)H + h) * D) * Tp + t0;
  for (int item = tid; item < D * 8; item += 256) {
    const int d = item >> 3, tc = item & 7;
    u16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = tile[(tc * 8 + e) * LDT + (d ^ (((tc * 2) & CMASK) << 3))];
    *reinterpret_cast<u16x8_t*>(out + (int64_t)d * Tp + tc * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------
// generic transpose: in [R, C] (ld) -> out [C, Rp] (ldo), rows R..Rp-1 written as zeros
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, int64_t ldi,
                                                        bf16_t* __restrict__ out, int64_t ldo, int R,
                                                        int C) {
  constexpr int LDT = 64 + 8;
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * LDT];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  for (int item = tid; item < 512; item += 256) {
    const int r = item >> 3, cc = item & 7;
    u16x8_t v;
    if (r0 + r < R && c0 + cc * 8 < C) {
      v = *reinterpret_cast<const u16x8_t*>(in + (int64_t)(r0 + r) * ldi + c0 + cc * 8);
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = 0;
    }
    *reinterpret_cast<u16x8_t*>(&tile[r * LDT + cc * 8]) = v;
  }
  __syncthreads();
  for (int item = tid; item < 512; item += 256) {
    const int c = item >> 3, rc = item & 7;
    if (c0 + c >= C) continue;
    u16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = tile[(rc * 8 + e) * LDT + c];
    *reinterpret_cast<u16x8_t*>(out + (int64_t)(c0 + c) * ldo + r0 + rc * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------
// SwiGLU
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, int64_t ldgu,
                                                         bf16_t* __restrict__ h, int64_t ldh,
                                                         int64_t M, int F) {
  const int nch = F >> 3;
  const int64_t total = M * nch;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / nch;
    const int c = (int)(i % nch);
    // (streamed once: non-temporal loads keep the 675 MB of [gate | up] from sweeping the L2 / Infinity Cache)
    const u16x8_t g = __builtin_nontemporal_load(reinterpret_cast<const u16x8_t*>(gu + m * ldgu + c * 8));
    const u16x8_t u = __builtin_nontemporal_load(reinterpret_cast<const u16x8_t*>(gu + m * ldgu + F + c * 8));
    u16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float gf = bf2f(g[e]);
      o[e] = f2bf(gf * sigmoid_fast(gf) * bf2f(u[e]));
    }
    *reinterpret_cast<u16x8_t*>(h + m * ldh + c * 8) = o;
  }
}

// IL: the forward's [gate | up] was written by the fused gate|up product (slam_gemm_swiglu_bf16_nt) in blocks of
// [64 gate columns | the matching 64 up columns]; dgate_up always leaves in the plain [dgate | dup] layout (the next product's
// weight W^T is stored that way).
template <bool IL>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ gu, int64_t ldgu,
                                                         const bf16_t* __restrict__ dh, int64_t lddh,
                                                         bf16_t* __restrict__ dgu, int64_t lddgu,
                                                         int64_t M, int F) {
  const int nch = F >> 3;
  const int64_t total = M * nch;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / nch;
    const int c = (int)(i % nch);
    const int gcol = IL ? ((c >> 3) << 7) + ((c & 7) << 3) : c * 8;      // column of gate[c * 8] in gu
    const int ucol = IL ? gcol + 64 : F + c * 8;
    const u16x8_t g = __builtin_nontemporal_load(reinterpret_cast<const u16x8_t*>(gu + m * ldgu + gcol));
    const u16x8_t u = __builtin_nontemporal_load(reinterpret_cast<const u16x8_t*>(gu + m * ldgu + ucol));
    const u16x8_t d = __builtin_nontemporal_load(reinterpret_cast<const u16x8_t*>(dh + m * lddh + c * 8));
    u16x8_t og, ou;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float gf = bf2f(g[e]), uf = bf2f(u[e]), df = bf2f(d[e]);
      const float sg = sigmoid_fast(gf);
      const float silu = gf * sg;
      og[e] = f2bf(df * uf * sg * (1.0f + gf * (1.0f - sg)));
      ou[e] = f2bf(df * silu);
    }
    *reinterpret_cast<u16x8_t*>(dgu + m * lddgu + c * 8) = og;
    *reinterpret_cast<u16x8_t*>(dgu + m * lddgu + F + c * 8) = ou;
  }
}

// ------------------------------------------------------------------------------------------
// conv1d(k=3, pad=1, stride s) im2col: in [B, Tin, C] -> out [B*Tout, Kp], col = j*C + c
// ------------------------------------------------------------------------------------------
// 8 output columns (16 bytes) per thread; needs C % 8 == 0 (a chunk never straddles two taps) and Kp % 8 == 0.
// n_valid (nullable, [B]): input frames at or beyond n_valid[b] read as zero -- a clip that is shorter than the batch's
// padded length then sees exactly the zero padding it would see alone (ragged encoder), whatever sits in the pad rows.
template <typename TIN>
__global__ __launch_bounds__(256) void im2col_k3_kernel(const TIN* __restrict__ in, bf16_t* __restrict__ out,
                                                        int B, int Tin, int Tout, int C, int Kp,
                                                        int stride, const int* __restrict__ n_valid) {
  const int KC = Kp / 8;
  const int64_t total = (int64_t)B * Tout * KC;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int col = (int)(i % KC) * 8;
    const int64_t row = i / KC;
    const int t = (int)(row % Tout);
    const int b = (int)(row / Tout);
    u16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = 0;
    if (col < 3 * C) {
      const int j = col / C, c = col % C;
      const int ti = t * stride + j - 1;
      const int lim = n_valid ? min(Tin, n_valid[b]) : Tin;
      if (ti >= 0 && ti < lim) {
        if constexpr (sizeof(TIN) == 4) {
          const float4 a = *reinterpret_cast<const float4*>(in + ((int64_t)b * Tin + ti) * C + c);
          const float4 d = *reinterpret_cast<const float4*>(in + ((int64_t)b * Tin + ti) * C + c + 4);
          o[0] = f2bf(a.x); o[1] = f2bf(a.y); o[2] = f2bf(a.z); o[3] = f2bf(a.w);
          o[4] = f2bf(d.x); o[5] = f2bf(d.y); o[6] = f2bf(d.z); o[7] = f2bf(d.w);
        } else {
          o = *reinterpret_cast<const u16x8_t*>(in + ((int64_t)b * Tin + ti) * C + c);
        }
      }
    }
    *reinterpret_cast<u16x8_t*>(out + row * Kp + col) = o;
  }
}

// adjoint of im2col_k3 (unfrozen-encoder training: dL/d(conv input) from dL/d(cols) = dz . W): gather form, no atomics --
//   dx[b, t, c] = sum_j dcols[b, o, j*C + c]  over the (o, j) with stride*o + j - 1 == t, 0 <= o < Tout
// (at most 3 terms for stride 1, 2 for stride 2); 8 channels per thread, fp32 sum, one bf16 rounding.
__global__ __launch_bounds__(256) void col2im_k3_kernel(const bf16_t* __restrict__ dcols, int64_t ldc, bf16_t* __restrict__ dx,
                                                        int B, int Tin, int Tout, int C, int stride) {
  const int CC = C / 8;
  const int64_t total = (int64_t)B * Tin * CC;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % CC) * 8;
    const int64_t row = i / CC;
    const int t = (int)(row % Tin);
    const int b = (int)(row / Tin);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int num = t + 1 - j;
      if (num < 0 || num % stride) continue;
      const int o = num / stride;
      if (o >= Tout) continue;
      const u16x8_t v = *reinterpret_cast<const u16x8_t*>(dcols + ((int64_t)b * Tout + o) * ldc + j * C + c);
#pragma unroll
      for (int e = 0; e < 8; e++) acc[e] += bf2f(v[e]);
    }
    u16x8_t o8;
#pragma unroll
    for (int e = 0; e < 8; e++) o8[e] = f2bf(acc[e]);
    *reinterpret_cast<u16x8_t*>(dx + row * C + c) = o8;
  }
}

// adjoint of the general im2col (conv feature extractor of HuBERT / WavLM, unfrozen-encoder training): kernel k, stride s, no padding --
//   dx[b, t, c] = sum_j dcols[b, o, j*C + c]  over the taps j < k with t - j = s * o, 0 <= o < Tout   (at most ceil(k / s) terms)
__global__ __launch_bounds__(256) void col2im_kernel(const bf16_t* __restrict__ dcols, int64_t ldc, bf16_t* __restrict__ dx, int B, int Tin,
                                                     int Tout, int C, int k, int stride) {
  const int CC = C / 8;
  const int64_t total = (int64_t)B * Tin * CC;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % CC) * 8;
    const int64_t row = i / CC;
    const int t = (int)(row % Tin);
    const int b = (int)(row / Tin);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = 0.f;
    for (int j = t % stride; j < k; j += stride) {   // t - j must be a multiple of the stride
      const int num = t - j;
      if (num < 0) break;
      const int o = num / stride;
      if (o >= Tout) continue;
      const u16x8_t v = *reinterpret_cast<const u16x8_t*>(dcols + ((int64_t)b * Tout + o) * ldc + j * C + c);
#pragma unroll
      for (int e = 0; e < 8; e++) acc[e] += bf2f(v[e]);
    }
    u16x8_t o8;
#pragma unroll
    for (int e = 0; e < 8; e++) o8[e] = f2bf(acc[e]);
    *reinterpret_cast<u16x8_t*>(dx + row * C + c) = o8;
  }
}

// dst[r, 0:width) = src[idx[r] * src_stride + 0:width)  (16 bytes per thread; idx[r] < 0 -> zeros).  One kernel for
// packing valid rows out of a padded batch, un-packing (inverse index, pad rows zero) and the projector's k-frame windows
// over a packed encoder output (width = k*d > src_stride = d: k consecutive rows are one contiguous window).
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ src, int64_t src_stride,
                                                          const int* __restrict__ idx, bf16_t* __restrict__ dst,
                                                          int64_t ld_dst, int64_t n, int width) {
  const int WC = width / 8;
  const int64_t total = n * WC;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / WC;
    const int c = (int)(i % WC) * 8;
    const int sidx = idx[r];
    u16x8_t v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = 0;
    if (sidx >= 0) v = *reinterpret_cast<const u16x8_t*>(src + (int64_t)sidx * src_stride + c);
    *reinterpret_cast<u16x8_t*>(dst + r * ld_dst + c) = v;
  }
}

// ------------------------------------------------------------------------------------------
// general conv1d im2col (HuBERT feature encoder k=10/3/2, stride 5/2, no padding; positional conv k=128, groups=16:
// fairseq/HF HubertModel reached from src/slam_llm/models/slam_model.py:335-341; conv spec in-repo at
// src/slam_llm/models/wavlm/WavLM.py:173,378-505).
// in: rows (b*Tin + t) with row stride ld_in, channels [c0, c0+C) -> out [B*Tout, Kp], col = j*C + c, zero padded.
// 16-byte path when the input is bf16 and C, c0, ld_in are multiples of 8.
// ------------------------------------------------------------------------------------------
template <typename TIN>
__global__ __launch_bounds__(256) void im2col_gen_kernel(const TIN* __restrict__ in, int64_t ld_in, int c0,
                                                         bf16_t* __restrict__ out, int B, int Tin, int Tout,
                                                         int C, int Kp, int k, int stride, int pad) {
  const int64_t total = (int64_t)B * Tout * Kp;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int col = (int)(i % Kp);
    const int64_t row = i / Kp;
    const int t = (int)(row % Tout);
    const int b = (int)(row / Tout);
    float v = 0.f;
    if (col < k * C) {
      const int j = col / C, c = col % C;
      const int ti = t * stride + j - pad;
      if (ti >= 0 && ti < Tin) {
        const TIN x = in[((int64_t)b * Tin + ti) * ld_in + c0 + c];
        if constexpr (sizeof(TIN) == 4) v = x; else v = bf2f(x);
      }
    }
    out[i] = f2bf(v);
  }
}

__global__ __launch_bounds__(256) void im2col_vec_kernel(const bf16_t* __restrict__ in, int64_t ld_in, int c0,
                                                         bf16_t* __restrict__ out, int B, int Tin, int Tout,
                                                         int C, int Kp, int k, int stride, int pad) {
  const int kpc = Kp >> 3, cc = C >> 3;
  const int64_t total = (int64_t)B * Tout * kpc;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % kpc);
    const int64_t row = i / kpc;
    const int t = (int)(row % Tout);
    const int b = (int)(row / Tout);
    u16x8_t v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = 0;
    if (ch < k * cc) {
      const int j = ch / cc, c8 = ch % cc;
      const int ti = t * stride + j - pad;
      if (ti >= 0 && ti < Tin)
        v = *reinterpret_cast<const u16x8_t*>(in + ((int64_t)b * Tin + ti) * ld_in + c0 + c8 * 8);
    }
    *reinterpret_cast<u16x8_t*>(out + row * Kp + ch * 8) = v;
  }
}

// ------------------------------------------------------------------------------------------
// fp32 -> bf16 cast
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in,
                                                            bf16_t* __restrict__ out, int64_t n) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = f2bf(in[i]);
}

__global__ __launch_bounds__(256) void add_bf16_kernel(bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ b,
                                                       int64_t ldb, int64_t M, int N) {
  const int nch = N >> 3;
  const int64_t total = M * nch;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / nch;
    const int c = (int)(i % nch);
    u16x8_t x = *reinterpret_cast<const u16x8_t*>(a + m * lda + c * 8);
    const u16x8_t y = *reinterpret_cast<const u16x8_t*>(b + m * ldb + c * 8);
#pragma unroll
    for (int e = 0; e < 8; e++) x[e] = f2bf(bf2f(x[e]) + bf2f(y[e]));
    *reinterpret_cast<u16x8_t*>(a + m * lda + c * 8) = x;
  }
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out,
                                                            int64_t n, int accumulate) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = accumulate ? out[i] + bf2f(in[i]) : bf2f(in[i]);
}

// ------------------------------------------------------------------------------------------
// modality spans + embed/splice
// ------------------------------------------------------------------------------------------
// spans[b] = (start, len): start = first True of modality_mask[b,:] (0 if none),
// len = min(sum(mask), Ta)   (slam_model.py:383-384)
__global__ __launch_bounds__(64) void modality_spans_kernel(const uint8_t* __restrict__ mask, int T,
                                                            int Ta, int* __restrict__ spans) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int first = T, cnt = 0;
  for (int t = lane; t < T; t += 64) {
    if (mask[(int64_t)b * T + t]) {
      cnt++;
      first = min(first, t);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    first = min(first, __shfl_xor(first, o, 64));
  }
  if (lane == 0) {
    spans[2 * b] = (first == T) ? 0 : first;
    spans[2 * b + 1] = min(cnt, Ta);
  }
}

__global__ __launch_bounds__(256) void embed_splice_kernel(
    int64_t* __restrict__ ids, const uint8_t* __restrict__ mask, const int* __restrict__ spans,
    const bf16_t* __restrict__ E, const bf16_t* __restrict__ enc, int64_t ldenc, bf16_t* __restrict__ out,
    int64_t ldo, int T, int Ta, int d, int64_t vocab) {
  const int64_t row = blockIdx.x;  // b*T + t
  const int b = (int)(row / T), t = (int)(row % T);
  const int start = spans[2 * b], len = spans[2 * b + 1];
  const bool in_span = (t >= start) && (t < start + len);
  const bool m = mask[row] != 0;
  int64_t id = ids[row];
  if (id == -1) {
    id = 0;
    if (threadIdx.x == 0) ids[row] = 0;  // the reference mutates input_ids in place (slam_model.py:371)
  }
  if (id < 0) id = 0;
  if (id >= vocab) id = vocab - 1;
  const bf16_t* er = E + id * (int64_t)d;
  const bf16_t* ar = enc + ((int64_t)b * Ta + (t - start)) * ldenc;
  bf16_t* orow = out + row * ldo;
  for (int c = threadIdx.x; c < (d >> 3); c += 256) {
    u16x8_t o;
    if (in_span && m) {
      o = *reinterpret_cast<const u16x8_t*>(ar + c * 8);
    } else if (!in_span && !m) {
      o = *reinterpret_cast<const u16x8_t*>(er + c * 8);
    } else if (in_span && !m) {  // encoder_outs_pad + embeds * 1  (never happens for contiguous masks)
      const u16x8_t a = *reinterpret_cast<const u16x8_t*>(ar + c * 8);
      const u16x8_t e8 = *reinterpret_cast<const u16x8_t*>(er + c * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = f2bf(bf2f(a[e]) + bf2f(e8[e]));
    } else {  // masked but beyond the (clamped) span: zero embedding
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = 0;
    }
    *reinterpret_cast<u16x8_t*>(orow + c * 8) = o;
  }
}

// d_enc[b, j, :] = j < len_b ? dX[b, start_b + j, :] : 0
__global__ __launch_bounds__(256) void embed_splice_bwd_kernel(const int* __restrict__ spans,
                                                               const bf16_t* __restrict__ dX, int64_t lddx,
                                                               bf16_t* __restrict__ denc, int64_t ldde,
                                                               int T, int Ta, int d) {
  const int64_t row = blockIdx.x;  // b*Ta + j
  const int b = (int)(row / Ta), j = (int)(row % Ta);
  const int start = spans[2 * b], len = spans[2 * b + 1];
  bf16_t* orow = denc + row * ldde;
  const bf16_t* irow = dX + ((int64_t)b * T + start + j) * lddx;
  for (int c = threadIdx.x; c < (d >> 3); c += 256) {
    u16x8_t o;
    if (j < len) {
      o = *reinterpret_cast<const u16x8_t*>(irow + c * 8);
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = 0;
    }
    *reinterpret_cast<u16x8_t*>(orow + c * 8) = o;
  }
}


// ------------------------------------------------------------------------------------------
// ReLU backward (projector, src/slam_llm/models/projector.py:25): dh *= (h > 0), in place
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relu_bwd_kernel(bf16_t* __restrict__ dh, int64_t lddh,
                                                       const bf16_t* __restrict__ h, int64_t ldh,
                                                       int64_t M, int N) {
  const int nch = N >> 3;
  const int64_t total = M * nch;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / nch;
    const int c = (int)(i % nch);
    u16x8_t d = *reinterpret_cast<const u16x8_t*>(dh + m * lddh + c * 8);
    const u16x8_t a = *reinterpret_cast<const u16x8_t*>(h + m * ldh + c * 8);
#pragma unroll
    for (int e = 0; e < 8; e++) d[e] = (bf2f(a[e]) > 0.f) ? d[e] : (bf16_t)0;
    *reinterpret_cast<u16x8_t*>(dh + m * lddh + c * 8) = d;
  }
}

// ------------------------------------------------------------------------------------------
// LoRA B packing: dst[row, j] = bf16(scale * B[row, j]) and dstT[j, row] = same (peft scaling = alpha / r)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lora_pack_b_kernel(const float* __restrict__ B, float scale,
                                                          bf16_t* __restrict__ dst, int64_t ld_dst,
                                                          bf16_t* __restrict__ dstT, int64_t ld_dstT,
                                                          int64_t rows, int r) {
  const int64_t total = rows * r;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / r;
    const int j = (int)(i % r);
    const bf16_t v = f2bf(B[i] * scale);
    dst[row * ld_dst + j] = v;
    dstT[(int64_t)j * ld_dstT + row] = v;
  }
}

// ------------------------------------------------------------------------------------------
// column sum (bias gradients of the projector Linears): out[n] (+)= sum_m x[m, n], deterministic order
// one workgroup = 64 columns; thread (ty = tid/8, tx = tid%8) walks rows ty, ty+32, ... on 8 columns
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                     float* __restrict__ out, int64_t M, int N,
                                                     int accumulate) {
  __shared__ float red[32][65];
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + tx * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < N) {
    for (int64_t m = ty; m < M; m += 32) {
      const u16x8_t v = *reinterpret_cast<const u16x8_t*>(x + m * ldx + c0);
#pragma unroll
      for (int e = 0; e < 8; e++) acc[e] += bf2f(v[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; e++) red[ty][tx * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < N) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 32; r++) s += red[r][threadIdx.x];
      out[c] = accumulate ? out[c] + s : s;
    }
  }
}

// ------------------------------------------------------------------------------------------
// GroupNorm with one group per channel over TIME + exact GELU: the first conv layer of the "default" feature extractor (WavLM
// Base / Base+, HuBERT-base; src/slam_llm/models/wavlm/WavLM.py:428-441: Fp32GroupNorm(dim, dim, affine=True)).  x is the conv
// output as fp32 [B*T, C] (time rows), every (clip, channel) is normalised over its T rows.  Three passes, HBM-bound:
//   partial: per (clip, 256-row chunk) column sums of x and x^2 in fp32 (threads = channels: coalesced rows)
//   final:   the chunks' partials combined in fp64 -> mean, rstd per (clip, channel)
//   apply:   y = gelu((x - mean) * rstd * weight + bias) as bf16
// ------------------------------------------------------------------------------------------
constexpr int GN_ROWS = 256;
__global__ __launch_bounds__(256) void gn_time_partial_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ part,
                                                              int T, int C, int nch) {
  const int ch = blockIdx.x, b = blockIdx.y;
  const int r0 = ch * GN_ROWS, r1 = min(T, r0 + GN_ROWS);
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f, q = 0.f;
    const float* p = x + ((int64_t)b * T + r0) * ldx + c;
    for (int r = r0; r < r1; r++, p += ldx) {
      const float v = *p;
      s += v;
      q = fmaf(v, v, q);
    }
    float* o = part + ((int64_t)(b * nch + ch) * 2) * C + c;
    o[0] = s;
    o[C] = q;
  }
}
__global__ __launch_bounds__(256) void gn_time_final_kernel(const float* __restrict__ part, float* __restrict__ stats, int T, int C,
                                                            int nch, float eps) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int ch = 0; ch < nch; ch++) {
    const float* o = part + ((int64_t)(b * nch + ch) * 2) * C + c;
    s += (double)o[0];
    q += (double)o[C];
  }
  const double mean = s / T;
  const double var = fmax(q / T - mean * mean, 0.0);      // biased variance, as torch's group_norm
  stats[((int64_t)b * 2) * C + c] = (float)mean;
  stats[((int64_t)b * 2 + 1) * C + c] = (float)(1.0 / sqrt(var + (double)eps));
}
__global__ __launch_bounds__(256) void gn_time_apply_gelu_kernel(const float* __restrict__ x, int64_t ldx, bf16_t* __restrict__ y,
                                                                 int64_t ldy, const float* __restrict__ stats,
                                                                 const float* __restrict__ weight, const float* __restrict__ bias,
                                                                 int64_t BT, int T, int C) {
  const int nq = C >> 2;
  const int64_t total = BT * nq;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / nq;
    const int c = (int)(i % nq) * 4;
    const int b = (int)(m / T);
    const float4 v = *reinterpret_cast<const float4*>(x + m * ldx + c);
    const float4 mu = *reinterpret_cast<const float4*>(stats + ((int64_t)b * 2) * C + c);
    const float4 rs = *reinterpret_cast<const float4*>(stats + ((int64_t)b * 2 + 1) * C + c);
    const float4 w4 = *reinterpret_cast<const float4*>(weight + c);
    const float4 b4 = *reinterpret_cast<const float4*>(bias + c);
    const float z[4] = {fmaf((v.x - mu.x) * rs.x, w4.x, b4.x), fmaf((v.y - mu.y) * rs.y, w4.y, b4.y),
                        fmaf((v.z - mu.z) * rs.z, w4.z, b4.z), fmaf((v.w - mu.w) * rs.w, w4.w, b4.w)};
    float g[4];
#pragma unroll
    for (int e = 0; e < 4; e++) g[e] = 0.5f * z[e] * (1.0f + erff(z[e] * 0.70710678118654752440f));
    uint2 o;
    o.x = pack2bf(g[0], g[1]);
    o.y = pack2bf(g[2], g[3]);
    *reinterpret_cast<uint2*>(y + m * ldy + c) = o;
  }
}

// backward of the above (unfrozen base-geometry encoders): with xh = (x - mean) rstd, z = w xh + b, dz = dy gelu'(z):
//   dx = w rstd (dz - mean_t(dz) - xh mean_t(dz xh)),  dw = sum_{b,t} dz xh,  db = sum_{b,t} dz.   Same three passes.
__device__ __forceinline__ float gn_dz(float x, float mu, float rs, float w, float b, float dy, float* xh_out) {
  const float xh = (x - mu) * rs;
  const float z = fmaf(xh, w, b);
  const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * z * z);
  *xh_out = xh;
  return dy * fmaf(z, pdf, cdf);
}
__global__ __launch_bounds__(256) void gn_time_bwd_partial_kernel(const float* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy,
                                                                  int64_t lddy, const float* __restrict__ stats,
                                                                  const float* __restrict__ weight, const float* __restrict__ bias,
                                                                  float* __restrict__ part, int T, int C, int nch) {
  const int ch = blockIdx.x, b = blockIdx.y;
  const int r0 = ch * GN_ROWS, r1 = min(T, r0 + GN_ROWS);
  for (int c = threadIdx.x; c < C; c += 256) {
    const float mu = stats[((int64_t)b * 2) * C + c], rs = stats[((int64_t)b * 2 + 1) * C + c], w = weight[c], bb = bias[c];
    float s1 = 0.f, s2 = 0.f;
    for (int r = r0; r < r1; r++) {
      const int64_t m = (int64_t)b * T + r;
      float xh;
      const float dz = gn_dz(x[m * ldx + c], mu, rs, w, bb, bf2f(dy[m * lddy + c]), &xh);
      s1 += dz;
      s2 = fmaf(dz, xh, s2);
    }
    float* o = part + ((int64_t)(b * nch + ch) * 2) * C + c;
    o[0] = s1;
    o[C] = s2;
  }
}
__global__ __launch_bounds__(256) void gn_time_bwd_final_kernel(const float* __restrict__ part, float* __restrict__ means, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int B, int T, int C, int nch, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double t1 = 0.0, t2 = 0.0;
  for (int b = 0; b < B; b++) {
    double s1 = 0.0, s2 = 0.0;
    for (int ch = 0; ch < nch; ch++) {
      const float* o = part + ((int64_t)(b * nch + ch) * 2) * C + c;
      s1 += (double)o[0];
      s2 += (double)o[C];
    }
    means[((int64_t)b * 2) * C + c] = (float)(s1 / T);
    means[((int64_t)b * 2 + 1) * C + c] = (float)(s2 / T);
    t1 += s1;
    t2 += s2;
  }
  dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)t1;
  dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)t2;
}
__global__ __launch_bounds__(256) void gn_time_bwd_apply_kernel(const float* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy,
                                                                int64_t lddy, const float* __restrict__ stats, const float* __restrict__ means,
                                                                const float* __restrict__ weight, const float* __restrict__ bias,
                                                                bf16_t* __restrict__ dx, int64_t lddx, int64_t BT, int T, int C) {
  const int nq = C >> 2;
  const int64_t total = BT * nq;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / nq;
    const int c = (int)(i % nq) * 4;
    const int b = (int)(m / T);
    const float4 v = *reinterpret_cast<const float4*>(x + m * ldx + c);
    const uint2 d2 = *reinterpret_cast<const uint2*>(dy + m * lddy + c);
    const float xv[4] = {v.x, v.y, v.z, v.w};
    const float dv[4] = {bf2f((uint16_t)(d2.x & 0xffff)), bf2f((uint16_t)(d2.x >> 16)), bf2f((uint16_t)(d2.y & 0xffff)), bf2f((uint16_t)(d2.y >> 16))};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float mu = stats[((int64_t)b * 2) * C + c + e], rs = stats[((int64_t)b * 2 + 1) * C + c + e];
      const float w = weight[c + e];
      float xh;
      const float dz = gn_dz(xv[e], mu, rs, w, bias[c + e], dv[e], &xh);
      o[e] = w * rs * (dz - means[((int64_t)b * 2) * C + c + e] - xh * means[((int64_t)b * 2 + 1) * C + c + e]);
    }
    uint2 ov;
    ov.x = pack2bf(o[0], o[1]);
    ov.y = pack2bf(o[2], o[3]);
    *reinterpret_cast<uint2*>(dx + m * lddx + c) = ov;
  }
}

// ------------------------------------------------------------------------------------------
// exact GELU forward / backward (Q-Former feed-forward, HF Blip2QFormerIntermediate): y = x * Phi(x)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ z, int64_t ldz, bf16_t* __restrict__ y,
                                                       int64_t ldy, int64_t M, int N) {
  const int nch = N >> 3;
  const int64_t total = M * nch;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / nch;
    const int c = (int)(i % nch);
    const u16x8_t v = *reinterpret_cast<const u16x8_t*>(z + m * ldz + c * 8);
    u16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float x = bf2f(v[e]);
      o[e] = f2bf(0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)));
    }
    *reinterpret_cast<u16x8_t*>(y + m * ldy + c * 8) = o;
  }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ z, int64_t ldz,
                                                       const bf16_t* __restrict__ dy, int64_t lddy,
                                                       bf16_t* __restrict__ dz, int64_t lddz, int64_t M, int N) {
  const int nch = N >> 3;
  const int64_t total = M * nch;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / nch;
    const int c = (int)(i % nch);
    const u16x8_t v = *reinterpret_cast<const u16x8_t*>(z + m * ldz + c * 8);
    const u16x8_t g = *reinterpret_cast<const u16x8_t*>(dy + m * lddy + c * 8);
    u16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float x = bf2f(v[e]);
      const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
      o[e] = f2bf(bf2f(g[e]) * (cdf + x * pdf));
    }
    *reinterpret_cast<u16x8_t*>(dz + m * lddz + c * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------
// counter-based dropout (peft LoRA `lora_dropout`, active in train mode: SURVEY g10):
//   out (+)= keep(seed, offset + m*N + n) ? x / (1 - p) : 0
// The mask is a pure function of (seed, offset, element index), so the backward recomputes it instead of storing it.
// One splitmix64 hash serves 4 consecutive elements (16 random bits each, keep = bits >= round(p * 2^16)): the
// 64-bit multiplies made a hash per element VALU-bound (58 us for 11780 x 4096 instead of the 35 us of its HBM traffic).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dropout_kernel(const bf16_t* __restrict__ x, int64_t ldx, bf16_t* __restrict__ out,
                                                      int64_t ldo, int64_t M, int N, float inv_keep, unsigned thresh,
                                                      unsigned long long seed, unsigned long long offset, int accumulate,
                                                      const unsigned long long* salt) {
  seed = slam_salted(seed, salt);
  const int nch = N >> 3;
  const int64_t total = M * nch;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / nch;
    const int c = (int)(i % nch);
    const u16x8_t v = *reinterpret_cast<const u16x8_t*>(x + m * ldx + c * 8);
    u16x8_t o;
    if (accumulate) o = *reinterpret_cast<const u16x8_t*>(out + m * ldo + c * 8);
    const unsigned long long base = offset + (unsigned long long)m * (unsigned long long)N + (unsigned long long)c * 8ull;
    const unsigned keep8 = slam_keep8(seed, base, thresh);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const bool keep = (keep8 >> e) & 1u;
      const float val = keep ? bf2f(v[e]) * inv_keep : 0.f;
      o[e] = f2bf(accumulate ? bf2f(o[e]) + val : val);
    }
    *reinterpret_cast<u16x8_t*>(out + m * ldo + c * 8) = o;
  }
}

// WavLM gate of the relative position bias (modules.py:522-531): one thread per (row, head)
__global__ __launch_bounds__(256) void wavlm_gate_kernel(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const float* __restrict__ grep_a,
                                                         float* __restrict__ gate, int B, int T, int H, int Tp) {
  __shared__ float ws[8 * 64 + 8];
  for (int i = threadIdx.x; i < 8 * 64 + 8; i += 256) ws[i] = i < 512 ? w[i] : bias[i - 512];
  __syncthreads();
  const int64_t total = (int64_t)B * T * H;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int h = (int)(i % H);
    const int64_t m = i / H;
    const int t = (int)(m % T), b = (int)(m / T);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = ws[512 + j];
#pragma unroll
    for (int c8 = 0; c8 < 8; c8++) {
      const u16x8_t xv = *reinterpret_cast<const u16x8_t*>(x + m * ldx + h * 64 + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float xf = bf2f(xv[e]);
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = fmaf(xf, ws[j * 64 + c8 * 8 + e], v[j]);
      }
    }
    const float a = 1.f / (1.f + __expf(-(v[0] + v[1] + v[2] + v[3])));
    const float g = 1.f / (1.f + __expf(-(v[4] + v[5] + v[6] + v[7])));
    gate[((int64_t)b * H + h) * Tp + t] = a * (g * grep_a[h] - 1.f) + 2.f;
  }
}

// backward of wavlm_gate_kernel (unfrozen WavLM): gate = a (g A - 1) + 2, a = sigmoid(sum of grep_linear outputs 0..3), g = sigmoid(sum of
// outputs 4..7), A = grep_a[h].  One thread per (row, head), recomputing the forward:
//   dv[(m, h), j]  = dL/d(grep_linear output j)   (bf16 [M*H, 8]: dW = dv^T x_h as a tall-skinny gram, db = column sums -- both by the
//                                                   existing fixed-order kernels, nothing is reduced with atomics here)
//   da_term[m, h]  = dgate * a * g                 (bf16 [M, Hp]: d grep_a = its column sums)
//   dx[m, h*64 + c] = sum_j dv_j * w[j][c]          (bf16: the gate's contribution to dL/d(attention input))
__global__ __launch_bounds__(256) void wavlm_gate_bwd_kernel(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const float* __restrict__ grep_a,
                                                             const float* __restrict__ dgate, bf16_t* __restrict__ dv, bf16_t* __restrict__ da_term,
                                                             bf16_t* __restrict__ dx, int64_t lddx, int B, int T, int H, int Tp, int Hp) {
  __shared__ float ws[8 * 64 + 8 + 128];   // w, bias, then wa[c] = sum_{j<4} w[j][c] and wg[c] = sum_{j>=4} w[j][c]
  for (int i = threadIdx.x; i < 8 * 64 + 8; i += 256) ws[i] = i < 512 ? w[i] : bias[i - 512];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x & 63, j0 = (threadIdx.x >> 6) * 4;
    ws[520 + threadIdx.x] = ws[j0 * 64 + c] + ws[(j0 + 1) * 64 + c] + ws[(j0 + 2) * 64 + c] + ws[(j0 + 3) * 64 + c];
  }
  __syncthreads();
  const int64_t total = (int64_t)B * T * H;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int h = (int)(i % H);
    const int64_t m = i / H;
    const int t = (int)(m % T), b = (int)(m / T);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = ws[512 + j];
    u16x8_t xv[8];
#pragma unroll
    for (int c8 = 0; c8 < 8; c8++) {
      xv[c8] = *reinterpret_cast<const u16x8_t*>(x + m * ldx + h * 64 + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float xf = bf2f(xv[c8][e]);
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = fmaf(xf, ws[j * 64 + c8 * 8 + e], v[j]);
      }
    }
    const float a = 1.f / (1.f + __expf(-(v[0] + v[1] + v[2] + v[3])));
    const float g = 1.f / (1.f + __expf(-(v[4] + v[5] + v[6] + v[7])));
    const float A = grep_a[h];
    const float dg_ = dgate[((int64_t)b * H + h) * Tp + t];
    const float dsa = dg_ * (g * A - 1.f) * a * (1.f - a);
    const float dsg = dg_ * a * A * g * (1.f - g);
    u16x8_t o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = f2bf(j < 4 ? dsa : dsg);
    *reinterpret_cast<u16x8_t*>(dv + i * 8) = o;
    da_term[m * Hp + h] = f2bf(dg_ * a * g);
#pragma unroll
    for (int c8 = 0; c8 < 8; c8++) {
      u16x8_t d8;
#pragma unroll
      for (int e = 0; e < 8; e++) d8[e] = f2bf(fmaf(dsa, ws[520 + c8 * 8 + e], dsg * ws[584 + c8 * 8 + e]));
      *reinterpret_cast<u16x8_t*>(dx + m * lddx + h * 64 + c8 * 8) = d8;
    }
  }
}

// chain rule of nn.utils.weight_norm(dim = 2) on a [rows, K] view of the positional conv weight (rows = d * channels per group, K = taps):
// w[:, k] = g[k] v[:, k] / ||v[:, k]||  ->  dg[k] = <dw[:, k], v[:, k]> / n,  dv[:, k] = g[k] / n (dw[:, k] - v[:, k] <dw, v> / n^2).
// One workgroup per tap (fixed-order block reductions: bit-reproducible); the tensor is 8 M elements at WavLM-Large.
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v, const float* __restrict__ g,
                                                              float* __restrict__ dg, float* __restrict__ dv, int rows, int K, int accumulate) {
  __shared__ float red[2][256];
  const int k = blockIdx.x, tid = threadIdx.x;
  float svv = 0.f, sdv = 0.f;
  for (int r = tid; r < rows; r += 256) {
    const float vv = v[(int64_t)r * K + k], dd = dw[(int64_t)r * K + k];
    svv = fmaf(vv, vv, svv);
    sdv = fmaf(dd, vv, sdv);
  }
  red[0][tid] = svv;
  red[1][tid] = sdv;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      red[0][tid] += red[0][tid + o];
      red[1][tid] += red[1][tid + o];
    }
    __syncthreads();
  }
  svv = red[0][0];
  sdv = red[1][0];
  const float n = sqrtf(svv), gk = g[k];
  if (tid == 0) dg[k] = (accumulate ? dg[k] : 0.f) + sdv / n;
  const float c1 = gk / n, c2 = sdv / svv;
  for (int r = tid; r < rows; r += 256) {
    const int64_t i = (int64_t)r * K + k;
    const float val = c1 * (dw[i] - v[i] * c2);
    dv[i] = accumulate ? dv[i] + val : val;
  }
}

// d(relative_attention_bias.weight)[bucket, h] (+)= sum over the distances r that fall in the bucket of d_table[h][r]
// (the bias table is the embedding gathered per relative distance, modules.py:444-455): one thread per (bucket, head), fixed order
__global__ __launch_bounds__(256) void relpos_bucket_grad_kernel(const float* __restrict__ d_tab, int64_t ld, const int* __restrict__ buckets, int n,
                                                                 int H, int nb, float* __restrict__ out, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nb * H) return;
  const int bucket = i / H, h = i % H;
  float acc = 0.f;
  for (int r = 0; r < n; r++)
    if (buckets[r] == bucket) acc += d_tab[(int64_t)h * ld + r];
  out[i] = accumulate ? out[i] + acc : acc;
}

inline unsigned ew_grid(int64_t total_items) {
  int64_t g = cdiv64(total_items, 256);
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" int slam_head_rope_transpose(void* src, int64_t ld, int64_t col0, const float* cos_table,
                                        const float* sin_table, int inverse, void* dstT, int64_t B,
                                        int64_t T, int64_t Tp, int64_t H, int64_t D, const int32_t* positions,
                                        void* stream) {
  SLAM_CHECK_ARG(src, "slam_head_rope_transpose: null src");
  SLAM_CHECK_ARG(D == 64 || D == 128, "slam_head_rope_transpose: head_dim %ld unsupported (64|128)", (long)D);
  SLAM_CHECK_ARG((cos_table == nullptr) == (sin_table == nullptr), "slam_head_rope_transpose: cos/sin must both be set or both null");
  SLAM_CHECK_ARG(ld % 8 == 0 && col0 % 8 == 0, "slam_head_rope_transpose: ld/col0 must be multiples of 8");
  SLAM_CHECK_ARG(B > 0 && T > 0 && H > 0, "slam_head_rope_transpose: bad shape");
  SLAM_CHECK_ARG(!dstT || (Tp % 64 == 0 && Tp >= T), "slam_head_rope_transpose: Tp must be a multiple of 64 and >= T");
  if (!dstT && !cos_table) return 0;
  const int64_t tt = dstT ? Tp : ((T + 63) / 64) * 64;
  dim3 grid((unsigned)(tt / 64), (unsigned)H, (unsigned)B);
  const float sgn = inverse ? -1.f : 1.f;
  if (D == 64)
    hipLaunchKernelGGL(head_rope_transpose_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)src, ld, (int)col0, cos_table, sin_table, sgn, (bf16_t*)dstT, (int)T, (int)Tp, (int)H, positions);
  else
    hipLaunchKernelGGL(head_rope_transpose_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)src, ld, (int)col0, cos_table, sin_table, sgn, (bf16_t*)dstT, (int)T, (int)Tp, (int)H, positions);
  SLAM_CHECK_LAUNCH("slam_head_rope_transpose");
  return 0;
}

extern "C" int slam_transpose_bf16(const void* in, int64_t ldi, void* out, int64_t ldo, int64_t R,
                                   int64_t C, int64_t Rp, void* stream) {
  SLAM_CHECK_ARG(in && out, "slam_transpose_bf16: null pointer");
  SLAM_CHECK_ARG(R > 0 && C > 0 && C % 8 == 0 && ldi % 8 == 0, "slam_transpose_bf16: C and ldi must be multiples of 8");
  SLAM_CHECK_ARG(Rp % 64 == 0 && Rp >= R && ldo >= Rp && ldo % 8 == 0, "slam_transpose_bf16: Rp must be a multiple of 64, >= R, <= ldo");
  dim3 grid((unsigned)(Rp / 64), (unsigned)cdiv64(C, 64));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ldi,
                     (bf16_t*)out, ldo, (int)R, (int)C);
  SLAM_CHECK_LAUNCH("slam_transpose_bf16");
  return 0;
}

extern "C" int slam_swiglu_fwd(const void* gate_up, int64_t ldgu, void* h, int64_t ldh, int64_t M,
                               int64_t F, void* stream) {
  SLAM_CHECK_ARG(gate_up && h, "slam_swiglu_fwd: null pointer");
  SLAM_CHECK_ARG(M > 0 && F > 0 && F % 8 == 0 && ldgu % 8 == 0 && ldh % 8 == 0, "slam_swiglu_fwd: F/ld must be multiples of 8");
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(ew_grid(M * (F / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)gate_up, ldgu, (bf16_t*)h, ldh, M, (int)F);
  SLAM_CHECK_LAUNCH("slam_swiglu_fwd");
  return 0;
}

extern "C" int slam_swiglu_bwd(const void* gate_up, int64_t ldgu, const void* dh, int64_t lddh,
                               void* dgate_up, int64_t lddgu, int64_t M, int64_t F, int interleaved, void* stream) {
  SLAM_CHECK_ARG(gate_up && dh && dgate_up, "slam_swiglu_bwd: null pointer");
  SLAM_CHECK_ARG(M > 0 && F > 0 && F % 8 == 0 && ldgu % 8 == 0 && lddh % 8 == 0 && lddgu % 8 == 0,
                 "slam_swiglu_bwd: F/ld must be multiples of 8");
  SLAM_CHECK_ARG(!interleaved || F % 64 == 0, "slam_swiglu_bwd: the interleaved [gate64 | up64] layout needs F %% 64 == 0 (F=%ld)", (long)F);
  if (interleaved)
    hipLaunchKernelGGL(swiglu_bwd_kernel<true>, dim3(ew_grid(M * (F / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)gate_up, ldgu, (const bf16_t*)dh, lddh, (bf16_t*)dgate_up, lddgu, M, (int)F);
  else
    hipLaunchKernelGGL(swiglu_bwd_kernel<false>, dim3(ew_grid(M * (F / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)gate_up, ldgu, (const bf16_t*)dh, lddh, (bf16_t*)dgate_up, lddgu, M, (int)F);
  SLAM_CHECK_LAUNCH("slam_swiglu_bwd");
  return 0;
}

extern "C" int slam_conv1d_k3_im2col(const void* in, int in_dtype, void* out, int64_t B, int64_t Tin,
                                     int64_t C, int64_t stride, int64_t Kp, const int32_t* n_valid, void* stream) {
  SLAM_CHECK_ARG(in && out, "slam_conv1d_k3_im2col: null pointer");
  SLAM_CHECK_ARG(stride == 1 || stride == 2, "slam_conv1d_k3_im2col: stride %ld unsupported", (long)stride);
  SLAM_CHECK_ARG(Kp >= 3 * C, "slam_conv1d_k3_im2col: Kp=%ld < 3*C=%ld", (long)Kp, (long)(3 * C));
  SLAM_CHECK_ARG(B > 0 && Tin > 0 && C > 0, "slam_conv1d_k3_im2col: bad shape");
  SLAM_CHECK_ARG(C % 8 == 0 && Kp % 8 == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0,
                 "slam_conv1d_k3_im2col: C=%ld and Kp=%ld must be multiples of 8, buffers 16-byte aligned", (long)C, (long)Kp);
  const int64_t Tout = (Tin + 2 - 3) / stride + 1;
  const int64_t total = B * Tout * (Kp / 8);
  if (in_dtype == SLAM_F32)
    hipLaunchKernelGGL(im2col_k3_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)in, (bf16_t*)out, (int)B, (int)Tin, (int)Tout, (int)C, (int)Kp, (int)stride, n_valid);
  else if (in_dtype == SLAM_BF16)
    hipLaunchKernelGGL(im2col_k3_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, (bf16_t*)out, (int)B, (int)Tin, (int)Tout, (int)C, (int)Kp, (int)stride, n_valid);
  else {
    slam_set_error("slam_conv1d_k3_im2col: in_dtype %d unknown", in_dtype);
    return -1;
  }
  SLAM_CHECK_LAUNCH("slam_conv1d_k3_im2col");
  return 0;
}

extern "C" int slam_conv1d_k3_col2im(const void* dcols, int64_t ldc, void* dx, int64_t B, int64_t Tin, int64_t C, int64_t stride,
                                     void* stream) {
  SLAM_CHECK_ARG(dcols && dx, "slam_conv1d_k3_col2im: null pointer");
  SLAM_CHECK_ARG(B > 0 && Tin > 0 && C > 0 && C % 8 == 0 && ldc % 8 == 0 && ldc >= 3 * C && (stride == 1 || stride == 2),
                 "slam_conv1d_k3_col2im: bad shape (C, ldc multiples of 8, ldc >= 3C, stride 1|2)");
  const int64_t Tout = (Tin + 2 - 3) / stride + 1;
  hipLaunchKernelGGL(col2im_k3_kernel, dim3(ew_grid(B * Tin * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dcols, ldc, (bf16_t*)dx, (int)B, (int)Tin, (int)Tout, (int)C, (int)stride);
  SLAM_CHECK_LAUNCH("slam_conv1d_k3_col2im");
  return 0;
}

extern "C" int slam_conv1d_col2im(const void* dcols, int64_t ldc, void* dx, int64_t B, int64_t Tin, int64_t C, int64_t k, int64_t stride,
                                  void* stream) {
  SLAM_CHECK_ARG(dcols && dx, "slam_conv1d_col2im: null pointer");
  SLAM_CHECK_ARG(B > 0 && Tin >= k && C > 0 && C % 8 == 0 && ldc % 8 == 0 && ldc >= k * C && k >= 1 && stride >= 1 && B * Tin < (1ll << 31),
                 "slam_conv1d_col2im: bad shape (C, ldc multiples of 8, ldc >= k*C, Tin >= k)");
  const int64_t Tout = (Tin - k) / stride + 1;
  hipLaunchKernelGGL(col2im_kernel, dim3(ew_grid(B * Tin * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dcols, ldc,
                     (bf16_t*)dx, (int)B, (int)Tin, (int)Tout, (int)C, (int)k, (int)stride);
  SLAM_CHECK_LAUNCH("slam_conv1d_col2im");
  return 0;
}

extern "C" int slam_wavlm_gate(const void* x, int64_t ldx, const float* w, const float* bias, const float* grep_a, float* gate,
                               int64_t B, int64_t T, int64_t H, int64_t Tp, void* stream) {
  SLAM_CHECK_ARG(x && w && bias && grep_a && gate, "slam_wavlm_gate: null pointer");
  SLAM_CHECK_ARG(B > 0 && T > 0 && H > 0 && Tp >= T && ldx % 8 == 0 && ldx >= H * 64, "slam_wavlm_gate: bad shape (head_dim is 64)");
  hipLaunchKernelGGL(wavlm_gate_kernel, dim3(ew_grid(B * T * H)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, w, bias,
                     grep_a, gate, (int)B, (int)T, (int)H, (int)Tp);
  SLAM_CHECK_LAUNCH("slam_wavlm_gate");
  return 0;
}

extern "C" int slam_weight_norm_bwd(const float* dw, const float* v, const float* g, float* dg, float* dv, int64_t rows, int64_t K, int accumulate,
                                    void* stream) {
  SLAM_CHECK_ARG(dw && v && g && dg && dv && rows > 0 && K > 0 && K < 65536 && rows < (1ll << 31), "slam_weight_norm_bwd: bad arguments");
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3((unsigned)K), dim3(256), 0, (hipStream_t)stream, dw, v, g, dg, dv, (int)rows, (int)K, accumulate);
  SLAM_CHECK_LAUNCH("slam_weight_norm_bwd");
  return 0;
}

extern "C" int slam_relpos_bucket_grad(const float* d_table, int64_t ld, const int32_t* buckets, int64_t n, int64_t H, int64_t num_buckets,
                                       float* out, int accumulate, void* stream) {
  SLAM_CHECK_ARG(d_table && buckets && out && n > 0 && H > 0 && num_buckets > 0 && ld >= n, "slam_relpos_bucket_grad: bad arguments");
  hipLaunchKernelGGL(relpos_bucket_grad_kernel, dim3((unsigned)cdiv64(num_buckets * H, 256)), dim3(256), 0, (hipStream_t)stream, d_table, ld,
                     buckets, (int)n, (int)H, (int)num_buckets, out, accumulate);
  SLAM_CHECK_LAUNCH("slam_relpos_bucket_grad");
  return 0;
}

extern "C" int slam_wavlm_gate_bwd(const void* x, int64_t ldx, const float* w, const float* bias, const float* grep_a, const float* dgate,
                                   void* dv, void* da_term, void* dx, int64_t lddx, int64_t B, int64_t T, int64_t H, int64_t Tp, int64_t Hp,
                                   void* stream) {
  SLAM_CHECK_ARG(x && w && bias && grep_a && dgate && dv && da_term && dx, "slam_wavlm_gate_bwd: null pointer");
  SLAM_CHECK_ARG(B > 0 && T > 0 && H > 0 && Tp >= T && Hp >= H && ldx % 8 == 0 && lddx % 8 == 0 && ldx >= H * 64 && lddx >= H * 64 &&
                     ((uintptr_t)dv % 16) == 0 && ((uintptr_t)dx % 16) == 0,
                 "slam_wavlm_gate_bwd: bad shape / alignment (head_dim is 64)");
  hipLaunchKernelGGL(wavlm_gate_bwd_kernel, dim3(ew_grid(B * T * H)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, w, bias, grep_a,
                     dgate, (bf16_t*)dv, (bf16_t*)da_term, (bf16_t*)dx, lddx, (int)B, (int)T, (int)H, (int)Tp, (int)Hp);
  SLAM_CHECK_LAUNCH("slam_wavlm_gate_bwd");
  return 0;
}

extern "C" int64_t slam_groupnorm_time_workspace_bytes(int64_t B, int64_t T, int64_t C) {
  const int64_t nch = (T + GN_ROWS - 1) / GN_ROWS;
  return (B * nch * 2 * C + B * 2 * C) * (int64_t)sizeof(float);
}

extern "C" int slam_groupnorm_time_gelu(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t B, int64_t T, int64_t C,
                                        const float* weight, const float* bias, float eps, float* workspace, void* stream) {
  SLAM_CHECK_ARG(x && y && weight && bias && workspace, "slam_groupnorm_time_gelu: null pointer");
  SLAM_CHECK_ARG(B > 0 && T > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C && B < 65536 &&
                     T < (1ll << 31) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 8) == 0 && ((uintptr_t)workspace % 16) == 0,
                 "slam_groupnorm_time_gelu: bad shape / alignment (C, ldx, ldy multiples of 4; x 16-byte aligned)");
  const int nch = (int)((T + GN_ROWS - 1) / GN_ROWS);
  float* part = workspace;
  float* stats = workspace + B * nch * 2 * C;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_time_partial_kernel, dim3((unsigned)nch, (unsigned)B), dim3(256), 0, s, x, ldx, part, (int)T, (int)C, nch);
  hipLaunchKernelGGL(gn_time_final_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)B), dim3(256), 0, s, part, stats, (int)T, (int)C,
                     nch, eps);
  hipLaunchKernelGGL(gn_time_apply_gelu_kernel, dim3(ew_grid(B * T * (C / 4))), dim3(256), 0, s, x, ldx, (bf16_t*)y, ldy, stats, weight,
                     bias, B * T, (int)T, (int)C);
  SLAM_CHECK_LAUNCH("slam_groupnorm_time_gelu");
  return 0;
}

extern "C" int slam_groupnorm_time_gelu_bwd(const float* x, int64_t ldx, const void* dy, int64_t lddy, const float* stats, const float* weight,
                                            const float* bias, void* dx, int64_t lddx, float* dgamma, float* dbeta, int64_t B, int64_t T,
                                            int64_t C, int accumulate, float* workspace, void* stream) {
  SLAM_CHECK_ARG(x && dy && stats && weight && bias && dx && dgamma && dbeta && workspace, "slam_groupnorm_time_gelu_bwd: null pointer");
  SLAM_CHECK_ARG(B > 0 && T > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ldx >= C && lddy >= C &&
                     lddx >= C && B < 65536 && T < (1ll << 31) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 8) == 0 &&
                     ((uintptr_t)dx % 8) == 0 && ((uintptr_t)workspace % 16) == 0,
                 "slam_groupnorm_time_gelu_bwd: bad shape / alignment (C and the leading dimensions multiples of 4)");
  const int nch = (int)((T + GN_ROWS - 1) / GN_ROWS);
  float* part = workspace;
  float* means = workspace + B * nch * 2 * C;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_time_bwd_partial_kernel, dim3((unsigned)nch, (unsigned)B), dim3(256), 0, s, x, ldx, (const bf16_t*)dy, lddy, stats,
                     weight, bias, part, (int)T, (int)C, nch);
  hipLaunchKernelGGL(gn_time_bwd_final_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s, part, means, dgamma, dbeta, (int)B, (int)T,
                     (int)C, nch, accumulate);
  hipLaunchKernelGGL(gn_time_bwd_apply_kernel, dim3(ew_grid(B * T * (C / 4))), dim3(256), 0, s, x, ldx, (const bf16_t*)dy, lddy, stats, means,
                     weight, bias, (bf16_t*)dx, lddx, B * T, (int)T, (int)C);
  SLAM_CHECK_LAUNCH("slam_groupnorm_time_gelu_bwd");
  return 0;
}

extern "C" int slam_gather_rows_bf16(const void* src, int64_t src_stride, const int32_t* idx, void* dst, int64_t ld_dst,
                                     int64_t n, int64_t width, void* stream) {
  SLAM_CHECK_ARG(src && idx && dst, "slam_gather_rows_bf16: null pointer");
  SLAM_CHECK_ARG(n > 0 && width > 0 && width % 8 == 0 && src_stride % 8 == 0 && ld_dst % 8 == 0 && ld_dst >= width &&
                     ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
                 "slam_gather_rows_bf16: width / strides must be multiples of 8 elements, buffers 16-byte aligned");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ew_grid(n * (width / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, src_stride, idx, (bf16_t*)dst, ld_dst, n, (int)width);
  SLAM_CHECK_LAUNCH("slam_gather_rows_bf16");
  return 0;
}

extern "C" int slam_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
  SLAM_CHECK_ARG(in && out && n > 0, "slam_cast_f32_to_bf16: bad arguments");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, in,
                     (bf16_t*)out, n);
  SLAM_CHECK_LAUNCH("slam_cast_f32_to_bf16");
  return 0;
}

extern "C" int slam_embed_splice_fwd(int64_t* input_ids, const uint8_t* modality_mask,
                                     const void* embed_table, int64_t vocab, const void* enc,
                                     int64_t ldenc, void* out, int64_t ldo, int32_t* spans, int64_t B,
                                     int64_t T, int64_t Ta, int64_t d, void* stream) {
  SLAM_CHECK_ARG(input_ids && modality_mask && embed_table && enc && out && spans, "slam_embed_splice_fwd: null pointer");
  SLAM_CHECK_ARG(B > 0 && T > 0 && Ta > 0 && d > 0 && d % 8 == 0 && ldenc % 8 == 0 && ldo % 8 == 0,
                 "slam_embed_splice_fwd: bad shape (d, ld multiples of 8)");
  hipLaunchKernelGGL(modality_spans_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream,
                     modality_mask, (int)T, (int)Ta, spans);
  hipLaunchKernelGGL(embed_splice_kernel, dim3((unsigned)(B * T)), dim3(256), 0, (hipStream_t)stream,
                     input_ids, modality_mask, spans, (const bf16_t*)embed_table, (const bf16_t*)enc, ldenc,
                     (bf16_t*)out, ldo, (int)T, (int)Ta, (int)d, vocab);
  SLAM_CHECK_LAUNCH("slam_embed_splice_fwd");
  return 0;
}

extern "C" int slam_embed_splice_bwd(const int32_t* spans, const void* dX, int64_t lddx, void* denc,
                                     int64_t ldde, int64_t B, int64_t T, int64_t Ta, int64_t d,
                                     void* stream) {
  SLAM_CHECK_ARG(spans && dX && denc, "slam_embed_splice_bwd: null pointer");
  SLAM_CHECK_ARG(B > 0 && T > 0 && Ta > 0 && d % 8 == 0 && lddx % 8 == 0 && ldde % 8 == 0,
                 "slam_embed_splice_bwd: bad shape");
  hipLaunchKernelGGL(embed_splice_bwd_kernel, dim3((unsigned)(B * Ta)), dim3(256), 0, (hipStream_t)stream,
                     spans, (const bf16_t*)dX, lddx, (bf16_t*)denc, ldde, (int)T, (int)Ta, (int)d);
  SLAM_CHECK_LAUNCH("slam_embed_splice_bwd");
  return 0;
}

extern "C" int slam_relu_bwd(void* dh, int64_t lddh, const void* h, int64_t ldh, int64_t M, int64_t N,
                             void* stream) {
  SLAM_CHECK_ARG(dh && h, "slam_relu_bwd: null pointer");
  SLAM_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0 && lddh % 8 == 0 && ldh % 8 == 0, "slam_relu_bwd: N/ld must be multiples of 8");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid(M * (N / 8))), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)dh, lddh, (const bf16_t*)h, ldh, M, (int)N);
  SLAM_CHECK_LAUNCH("slam_relu_bwd");
  return 0;
}

extern "C" int slam_lora_pack_b(const float* B, float scale, void* dst, int64_t ld_dst, void* dstT,
                                int64_t ld_dstT, int64_t rows, int64_t r, void* stream) {
  SLAM_CHECK_ARG(B && dst && dstT && rows > 0 && r > 0, "slam_lora_pack_b: bad arguments");
  hipLaunchKernelGGL(lora_pack_b_kernel, dim3(ew_grid(rows * r)), dim3(256), 0, (hipStream_t)stream, B,
                     scale, (bf16_t*)dst, ld_dst, (bf16_t*)dstT, ld_dstT, rows, (int)r);
  SLAM_CHECK_LAUNCH("slam_lora_pack_b");
  return 0;
}

extern "C" int slam_colsum_bf16(const void* x, int64_t ldx, float* out, int64_t M, int64_t N,
                                int accumulate, void* stream) {
  SLAM_CHECK_ARG(x && out && M > 0 && N > 0 && N % 8 == 0 && ldx % 8 == 0, "slam_colsum_bf16: bad arguments (N, ld multiples of 8)");
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)cdiv64(N, 64)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ldx, out, M, (int)N, accumulate);
  SLAM_CHECK_LAUNCH("slam_colsum_bf16");
  return 0;
}

extern "C" int slam_conv1d_im2col(const void* in, int in_dtype, int64_t ld_in, int64_t c0, int64_t C, void* out,
                                  int64_t B, int64_t Tin, int64_t k, int64_t stride, int64_t pad, int64_t Kp,
                                  int64_t Tout_limit, void* stream) {
  SLAM_CHECK_ARG(in && out, "slam_conv1d_im2col: null pointer");
  SLAM_CHECK_ARG(B > 0 && Tin > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && c0 >= 0, "slam_conv1d_im2col: bad shape");
  SLAM_CHECK_ARG(Kp >= k * C && Kp % 8 == 0, "slam_conv1d_im2col: Kp=%ld must be >= k*C=%ld and a multiple of 8", (long)Kp, (long)(k * C));
  SLAM_CHECK_ARG(Tin + 2 * pad >= k, "slam_conv1d_im2col: input shorter than the kernel");
  int64_t Tout = (Tin + 2 * pad - k) / stride + 1;
  if (Tout_limit > 0 && Tout_limit < Tout) Tout = Tout_limit;  // e.g. even-kernel "same" padding drops the last frame
  hipStream_t s = (hipStream_t)stream;
  if (in_dtype == SLAM_BF16 && C % 8 == 0 && c0 % 8 == 0 && ld_in % 8 == 0) {
    hipLaunchKernelGGL(im2col_vec_kernel, dim3(ew_grid(B * Tout * (Kp / 8))), dim3(256), 0, s, (const bf16_t*)in, ld_in,
                       (int)c0, (bf16_t*)out, (int)B, (int)Tin, (int)Tout, (int)C, (int)Kp, (int)k, (int)stride, (int)pad);
  } else if (in_dtype == SLAM_BF16) {
    hipLaunchKernelGGL(im2col_gen_kernel<bf16_t>, dim3(ew_grid(B * Tout * Kp)), dim3(256), 0, s, (const bf16_t*)in, ld_in,
                       (int)c0, (bf16_t*)out, (int)B, (int)Tin, (int)Tout, (int)C, (int)Kp, (int)k, (int)stride, (int)pad);
  } else if (in_dtype == SLAM_F32) {
    hipLaunchKernelGGL(im2col_gen_kernel<float>, dim3(ew_grid(B * Tout * Kp)), dim3(256), 0, s, (const float*)in, ld_in,
                       (int)c0, (bf16_t*)out, (int)B, (int)Tin, (int)Tout, (int)C, (int)Kp, (int)k, (int)stride, (int)pad);
  } else {
    slam_set_error("slam_conv1d_im2col: in_dtype %d unknown", in_dtype);
    return -1;
  }
  SLAM_CHECK_LAUNCH("slam_conv1d_im2col");
  return 0;
}

extern "C" int slam_gelu_fwd(const void* z, int64_t ldz, void* y, int64_t ldy, int64_t M, int64_t N, void* stream) {
  SLAM_CHECK_ARG(z && y && M > 0 && N > 0 && N % 8 == 0 && ldz % 8 == 0 && ldy % 8 == 0, "slam_gelu_fwd: bad arguments");
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(ew_grid(M * (N / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, ldz,
                     (bf16_t*)y, ldy, M, (int)N);
  SLAM_CHECK_LAUNCH("slam_gelu_fwd");
  return 0;
}

extern "C" int slam_gelu_bwd(const void* z, int64_t ldz, const void* dy, int64_t lddy, void* dz, int64_t lddz,
                             int64_t M, int64_t N, void* stream) {
  SLAM_CHECK_ARG(z && dy && dz && M > 0 && N > 0 && N % 8 == 0 && ldz % 8 == 0 && lddy % 8 == 0 && lddz % 8 == 0,
                 "slam_gelu_bwd: bad arguments");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_grid(M * (N / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, ldz,
                     (const bf16_t*)dy, lddy, (bf16_t*)dz, lddz, M, (int)N);
  SLAM_CHECK_LAUNCH("slam_gelu_bwd");
  return 0;
}

extern "C" int slam_cast_bf16_to_f32(const void* in, float* out, int64_t n, int accumulate, void* stream) {
  SLAM_CHECK_ARG(in && out && n > 0, "slam_cast_bf16_to_f32: bad arguments");
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, out, n,
                     accumulate);
  SLAM_CHECK_LAUNCH("slam_cast_bf16_to_f32");
  return 0;
}

extern "C" int slam_add_bf16(void* a, int64_t lda, const void* b, int64_t ldb, int64_t M, int64_t N, void* stream) {
  SLAM_CHECK_ARG(a && b && M > 0 && N > 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "slam_add_bf16: bad arguments");
  hipLaunchKernelGGL(add_bf16_kernel, dim3(ew_grid(M * (N / 8))), dim3(256), 0, (hipStream_t)stream, (bf16_t*)a, lda,
                     (const bf16_t*)b, ldb, M, (int)N);
  SLAM_CHECK_LAUNCH("slam_add_bf16");
  return 0;
}

extern "C" int slam_dropout_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, int64_t M, int64_t N, float p,
                                 uint64_t seed, uint64_t offset, int accumulate, void* stream) {
  SLAM_CHECK_ARG(x && out && M > 0 && N > 0 && N % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0, "slam_dropout_bf16: bad arguments");
  SLAM_CHECK_ARG(p >= 0.f && p < 1.f, "slam_dropout_bf16: p=%f must be in [0, 1)", (double)p);
  SLAM_CHECK_ARG(offset % 8 == 0, "slam_dropout_bf16: offset must be a multiple of 8");
  const unsigned thresh = slam_drop_thresh16(p);
  hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(M * (N / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                     (bf16_t*)out, ldo, M, (int)N, 1.0f / (1.0f - p), thresh, (unsigned long long)seed,
                     (unsigned long long)offset, accumulate, g_slam_drop_salt);
  SLAM_CHECK_LAUNCH("slam_dropout_bf16");
  return 0;
}
