// Shifted masked cross-entropy + argmax accuracy over bf16 logits, and fused flat AdamW.
//
// Reference semantics:
//  * HF ForCausalLMLoss (transformers/loss/loss_utils.py:32-70): logits upcast to fp32, labels shifted by
//    one, mean CE over labels != -100 (ignore_index), per local batch.
//  * accuracy: argmax(logits)[:, :-1] == labels[:, 1:] over labels != -100
//    (src/slam_llm/models/slam_model.py:402-405, src/slam_llm/utils/metric.py:3-19).
//  * AdamW: torch.optim.AdamW as constructed at src/slam_llm/pipeline/finetune.py:247-251
//    (decoupled weight decay, bias correction, eps added after sqrt(v_hat)).
// The CE kernel never materialises fp32 logits; rows whose shifted label is ignored only get their gradient
// row zeroed.  Row losses are reduced by a single-block kernel in a fixed order -> bit-reproducible loss.
#include "common.h"

namespace {

// tgt[row=b*T+t] = t+1<T ? labels[b,t+1] : -100 ; n_valid = #(tgt != ignore)
__global__ __launch_bounds__(1024) void ce_targets_kernel(const int64_t* __restrict__ labels,
                                                          int32_t* __restrict__ tgt,
                                                          int32_t* __restrict__ n_valid, int64_t M, int T,
                                                          int ignore_index) {
  __shared__ int red[16];
  int cnt = 0;
  for (int64_t r = threadIdx.x; r < M; r += 1024) {
    const int t = (int)(r % T);
    int64_t l = (t + 1 < T) ? labels[r + 1] : (int64_t)ignore_index;
    if (l != ignore_index) cnt++;
    tgt[r] = (l == ignore_index) ? -1 : (int32_t)l;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int i = 0; i < 16; i++) s += red[i];
    *n_valid = s;
  }
}

struct MaxSum {
  float m, s;
  int idx;
};
__device__ __forceinline__ MaxSum ms_combine(MaxSum a, MaxSum b) {
  MaxSum r;
  if (b.m > a.m || (b.m == a.m && b.idx < a.idx)) {
    r.m = b.m; r.idx = b.idx;
  } else {
    r.m = a.m; r.idx = a.idx;
  }
  // (-inf, 0) is the identity element (lanes that saw no chunk); avoid exp(-inf - -inf) = NaN
  const float ea = (a.m == -INFINITY) ? 0.f : __expf(a.m - r.m);
  const float eb = (b.m == -INFINITY) ? 0.f : __expf(b.m - r.m);
  r.s = a.s * ea + b.s * eb;
  return r;
}

// one workgroup per logits row; logits are overwritten with dlogits when write_grad != 0
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(bf16_t* __restrict__ logits, int64_t ld,
                                                         const int32_t* __restrict__ tgt,
                                                         const int32_t* __restrict__ n_valid,
                                                         float* __restrict__ row_loss,
                                                         int32_t* __restrict__ row_correct, int V,
                                                         int write_grad) {
  __shared__ float sm[4], ss[4];
  __shared__ int si[4];
  const int64_t row = blockIdx.x;
  const int target = tgt[row];
  bf16_t* lr = logits + row * ld;
  const int nch = V >> 3;
  const int tid = threadIdx.x;
  if (target < 0) {
    if (tid == 0) {
      row_loss[row] = 0.f;
      row_correct[row] = 0;
    }
    if (write_grad) {
      u16x8_t z;
#pragma unroll
      for (int e = 0; e < 8; e++) z[e] = 0;
      for (int c = tid; c < nch; c += 256) *reinterpret_cast<u16x8_t*>(lr + c * 8) = z;
    }
    return;
  }
  MaxSum a;
  a.m = -INFINITY; a.s = 0.f; a.idx = 0x7fffffff;
  for (int c = tid; c < nch; c += 256) {
    const u16x8_t v = *reinterpret_cast<const u16x8_t*>(lr + c * 8);
    float f[8];
    float cm = -INFINITY;
    int ci = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      f[e] = bf2f(v[e]);
      if (f[e] > cm) { cm = f[e]; ci = e; }
    }
    float cs = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) cs += __expf(f[e] - cm);
    MaxSum b;
    b.m = cm; b.s = cs; b.idx = c * 8 + ci;
    a = ms_combine(a, b);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MaxSum b;
    b.m = __shfl_xor(a.m, o, 64);
    b.s = __shfl_xor(a.s, o, 64);
    b.idx = __shfl_xor(a.idx, o, 64);
    a = ms_combine(a, b);
  }
  if ((tid & 63) == 0) { sm[tid >> 6] = a.m; ss[tid >> 6] = a.s; si[tid >> 6] = a.idx; }
  __syncthreads();
  MaxSum t;
  t.m = sm[0]; t.s = ss[0]; t.idx = si[0];
#pragma unroll
  for (int i = 1; i < 4; i++) {
    MaxSum b;
    b.m = sm[i]; b.s = ss[i]; b.idx = si[i];
    t = ms_combine(t, b);
  }
  const float lse = t.m + __logf(t.s);
  if (tid == 0) {
    row_loss[row] = lse - bf2f(lr[target]);
    row_correct[row] = (t.idx == target) ? 1 : 0;
  }
  if (write_grad) {
    __syncthreads();  // the target logit has been read by thread 0 before it is overwritten
    const float inv_n = 1.0f / (float)max(*n_valid, 1);
    for (int c = tid; c < nch; c += 256) {
      const u16x8_t v = *reinterpret_cast<const u16x8_t*>(lr + c * 8);
      u16x8_t o;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float p = __expf(bf2f(v[e]) - lse);
        if (c * 8 + e == target) p -= 1.0f;
        o[e] = f2bf(p * inv_n);
      }
      *reinterpret_cast<u16x8_t*>(lr + c * 8) = o;
    }
  }
}

// out[0] = mean loss, out[1] = accuracy, fixed summation order
__global__ __launch_bounds__(1024) void ce_finalize_kernel(const float* __restrict__ row_loss,
                                                           const int32_t* __restrict__ row_correct,
                                                           const int32_t* __restrict__ n_valid, int64_t M,
                                                           float* __restrict__ out) {
  __shared__ float rl[16];
  __shared__ int rc[16];
  float l = 0.f;
  int c = 0;
  for (int64_t r = threadIdx.x; r < M; r += 1024) {
    l += row_loss[r];
    c += row_correct[r];
  }
  l = wave_sum(l);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) { rl[threadIdx.x >> 6] = l; rc[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float L = 0.f;
    int C = 0;
    for (int i = 0; i < 16; i++) { L += rl[i]; C += rc[i]; }
    const float n = (float)(*n_valid);
    out[0] = L / n;   // n == 0 -> NaN, same as torch's mean over an empty selection
    out[1] = (float)C / n;
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ p_bf16, int64_t n, float lr,
                                                    float beta1, float beta2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, float gscale) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    float pi = p[i];
    pi *= (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (p_bf16) p_bf16[i] = f2bf(pi);
  }
}

}  // namespace

extern "C" int slam_ce_targets(const int64_t* labels, int32_t* targets, int32_t* n_valid, int64_t B,
                               int64_t T, int64_t ignore_index, void* stream) {
  SLAM_CHECK_ARG(labels && targets && n_valid && B > 0 && T > 0, "slam_ce_targets: bad arguments");
  hipLaunchKernelGGL(ce_targets_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, labels, targets,
                     n_valid, B * T, (int)T, (int)ignore_index);
  SLAM_CHECK_LAUNCH("slam_ce_targets");
  return 0;
}

extern "C" int slam_ce_fwd_bwd(void* logits, int64_t ld, const int32_t* targets, const int32_t* n_valid,
                               float* row_loss, int32_t* row_correct, int64_t rows, int64_t V,
                               int write_grad, void* stream) {
  SLAM_CHECK_ARG(logits && targets && n_valid && row_loss && row_correct, "slam_ce_fwd_bwd: null pointer");
  SLAM_CHECK_ARG(rows > 0 && V > 0 && V % 8 == 0 && ld % 8 == 0 && ld >= V, "slam_ce_fwd_bwd: V and ld must be multiples of 8");
  hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)logits, ld, targets, n_valid, row_loss, row_correct, (int)V, write_grad);
  SLAM_CHECK_LAUNCH("slam_ce_fwd_bwd");
  return 0;
}

extern "C" int slam_ce_finalize(const float* row_loss, const int32_t* row_correct, const int32_t* n_valid,
                                int64_t rows, float* out2, void* stream) {
  SLAM_CHECK_ARG(row_loss && row_correct && n_valid && out2 && rows > 0, "slam_ce_finalize: bad arguments");
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, row_loss,
                     row_correct, n_valid, rows, out2);
  SLAM_CHECK_LAUNCH("slam_ce_finalize");
  return 0;
}

extern "C" int slam_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                               void* param_bf16, int64_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step, float grad_scale, void* stream) {
  SLAM_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0, "slam_adamw_step: bad arguments");
  SLAM_CHECK_ARG(step >= 1, "slam_adamw_step: step must be >= 1 (got %ld)", (long)step);
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2 = 1.0f - powf(beta2, (float)step);
  int64_t g = cdiv64(n, 256);
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, param, grad,
                     exp_avg, exp_avg_sq, (bf16_t*)param_bf16, n, lr, beta1, beta2, eps, weight_decay, bc1,
                     sqrtf(bc2), grad_scale);
  SLAM_CHECK_LAUNCH("slam_adamw_step");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Tall-skinny gram product for the LoRA gradients (peft Linear backward: dA = (dy.sB)^T x, dB = s dy^T (xA^T)):
//   out[r, c] (+)= alpha * sum_m S[m, r] * X[m, c]       S: [M, R] bf16 with R in {8,16,32,64},  X: [M, C] bf16
// The reduction runs over the token dimension M (~12 k) while R is tiny, so this is an HBM-bound stream over X,
// not an MFMA problem: lane = 8 consecutive columns of X (16-byte loads), wave w = R/4 rows of the output, fp32
// FMA accumulation; M is split over gridDim.y and the partials are reduced in a fixed order by a second kernel
// (bit-reproducible gradients).  Replaces three bf16 transposes + three starved-grid GEMMs per adapted projection.
// ------------------------------------------------------------------------------------------------------------
namespace {

template <int RW>
__global__ __launch_bounds__(256) void skinny_gram_kernel(const bf16_t* __restrict__ S, int64_t lds_,
                                                          const bf16_t* __restrict__ X, int64_t ldx,
                                                          float* __restrict__ ws, int M, int R, int C,
                                                          int rows_per_split, unsigned thresh16,
                                                          unsigned long long seed, unsigned long long offset) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 512 + lane * 8;
  const int r0 = wave * RW;
  const int m0 = blockIdx.y * rows_per_split;
  const int m1 = min(M, m0 + rows_per_split);
  float acc[RW][8];
#pragma unroll
  for (int i = 0; i < RW; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) acc[i][e] = 0.f;
  if (c0 < C) {
    typedef __attribute__((ext_vector_type(RW))) unsigned short svec_t;  // RW bf16 of S in one load
    constexpr int U = 4;                                                 // rows in flight per iteration
    int m = m0;
    for (; m + U <= m1; m += U) {
      u16x8_t xv[U];
      svec_t sv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        xv[u] = *reinterpret_cast<const u16x8_t*>(X + (int64_t)(m + u) * ldx + c0);
        sv[u] = *reinterpret_cast<const svec_t*>(S + (int64_t)(m + u) * lds_ + r0);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        float xf[8];
        // X = dropout(x) recomputed from the counter-based mask (the 1/(1-p) scale is folded into alpha by the host)
        const unsigned keep = thresh16 ? slam_keep8(seed, offset + (unsigned long long)(m + u) * (unsigned long long)C + c0, thresh16) : 0xFFu;
#pragma unroll
        for (int e = 0; e < 8; e++) xf[e] = ((keep >> e) & 1u) ? bf2f(xv[u][e]) : 0.f;
#pragma unroll
        for (int i = 0; i < RW; i++) {
          const float sf = bf2f(sv[u][i]);
#pragma unroll
          for (int e = 0; e < 8; e++) acc[i][e] = fmaf(sf, xf[e], acc[i][e]);
        }
      }
    }
    for (; m < m1; m++) {
      const u16x8_t xv = *reinterpret_cast<const u16x8_t*>(X + (int64_t)m * ldx + c0);
      const svec_t sv = *reinterpret_cast<const svec_t*>(S + (int64_t)m * lds_ + r0);
      const unsigned keep = thresh16 ? slam_keep8(seed, offset + (unsigned long long)m * (unsigned long long)C + c0, thresh16) : 0xFFu;
#pragma unroll
      for (int i = 0; i < RW; i++) {
        const float sf = bf2f(sv[i]);
#pragma unroll
        for (int e = 0; e < 8; e++) acc[i][e] = fmaf(sf, ((keep >> e) & 1u) ? bf2f(xv[e]) : 0.f, acc[i][e]);
      }
    }
    float* w = ws + ((int64_t)blockIdx.y * R + r0) * C + c0;
#pragma unroll
    for (int i = 0; i < RW; i++) {
      *reinterpret_cast<float4*>(w + (int64_t)i * C) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
      *reinterpret_cast<float4*>(w + (int64_t)i * C + 4) = make_float4(acc[i][4], acc[i][5], acc[i][6], acc[i][7]);
    }
  }
}

__global__ __launch_bounds__(256) void skinny_gram_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                                 int64_t ld_r, int64_t ld_c, int R, int C, int nsplit,
                                                                 float alpha, int accumulate) {
  const int64_t total = (int64_t)R * C;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int k = 0; k < nsplit; k++) s += ws[(int64_t)k * total + i];
    const int r = (int)(i / C), c = (int)(i % C);
    float* o = out + r * ld_r + c * ld_c;
    *o = accumulate ? (*o + alpha * s) : alpha * s;
  }
}

// ------------------------------------------------------------------------------------------------------------
// LoRA first hop  u[M, R] = dropout(x)[M, K] . A[R, K]^T   (peft: lora_A(lora_dropout(x)), R = sum of the ranks of the
// adapters sharing x, <= 64).  HBM-bound on x: a 128-row MFMA GEMM tile gives only M/128 workgroups (93 for the C3
// batch) -- here a workgroup owns 32 rows and its 4 waves split K (fixed-order LDS reduction), the dropout mask is
// recomputed in registers, so x is read once and no dropout(x) copy is ever written.
// ------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void lora_a_fwd_kernel(const bf16_t* __restrict__ X, int64_t ldx,
                                                         const bf16_t* __restrict__ A, int64_t lda,
                                                         bf16_t* __restrict__ U, int64_t ldu, int M, int R, int K,
                                                         unsigned thresh16, float inv_keep, unsigned long long seed,
                                                         unsigned long long offset) {
  __shared__ float red[4][2 * NT][256 + 4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * 32;
  const int kw = (int)(((K / 64 + 3) / 4) * 64);
  const int k_begin = wave * kw, k_end = min(K, k_begin + kw);
  int mrow[2];
  const bf16_t* xp[2];
#pragma unroll
  for (int mt = 0; mt < 2; mt++) {
    mrow[mt] = min(m0 + mt * 16 + r, M - 1);
    xp[mt] = X + (int64_t)mrow[mt] * ldx + g * 16;
  }
  const bf16_t* ap[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) ap[nt] = A + (int64_t)min(nt * 16 + r, R - 1) * lda + g * 16;
  f32x4_t acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int k0 = k_begin; k0 < k_end; k0 += 64) {
    u16x8_t xf[2][2], af[NT][2];
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
      xf[mt][0] = *reinterpret_cast<const u16x8_t*>(xp[mt] + k0);
      xf[mt][1] = *reinterpret_cast<const u16x8_t*>(xp[mt] + k0 + 8);
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      af[nt][0] = *reinterpret_cast<const u16x8_t*>(ap[nt] + k0);
      af[nt][1] = *reinterpret_cast<const u16x8_t*>(ap[nt] + k0 + 8);
    }
    if (thresh16) {
#pragma unroll
      for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
          const unsigned keep = slam_keep8(seed, offset + (unsigned long long)mrow[mt] * (unsigned long long)K +
                                                     (unsigned long long)(k0 + g * 16 + hf * 8), thresh16);
#pragma unroll
          for (int e = 0; e < 8; e++)
            if (!((keep >> e) & 1u)) xf[mt][hf][e] = 0;
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[nt][0]),
                                                             __builtin_bit_cast(bf16x8_t, xf[mt][0]), acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[nt][1]),
                                                             __builtin_bit_cast(bf16x8_t, xf[mt][1]), acc[mt][nt], 0, 0, 0);
      }
  }
  // acc[mt][nt][i] = u[m0 + mt*16 + r][nt*16 + 4g + i]
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) *reinterpret_cast<f32x4_t*>(&red[wave][mt * NT + nt][lane * 4]) = acc[mt][nt];
  __syncthreads();
  for (int idx = tid; idx < 32 * NT * 4; idx += 256) {   // (row, 4 consecutive ranks)
    const int row = idx / (NT * 4), jq = idx % (NT * 4);
    const int mt = row >> 4, rr = row & 15, nt = jq >> 2, gg = jq & 3;
    const int m = m0 + row, j = nt * 16 + gg * 4;
    if (m >= M || j >= R) continue;
    const int src = (gg * 16 + rr) * 4;
    f32x4_t v = *reinterpret_cast<const f32x4_t*>(&red[0][mt * NT + nt][src]);
#pragma unroll
    for (int w = 1; w < 4; w++) {
      const f32x4_t t = *reinterpret_cast<const f32x4_t*>(&red[w][mt * NT + nt][src]);
#pragma unroll
      for (int i = 0; i < 4; i++) v[i] += t[i];
    }
    uint2 o;
    o.x = pack2bf(v[0] * inv_keep, v[1] * inv_keep);
    o.y = pack2bf(v[2] * inv_keep, v[3] * inv_keep);
    *reinterpret_cast<uint2*>(U + (int64_t)m * ldu + j) = o;
  }
}

}  // namespace

extern "C" int slam_lora_a_fwd(const void* X, int64_t ldx, const void* A, int64_t lda, void* U, int64_t ldu, int64_t M,
                               int64_t R, int64_t K, float drop_p, uint64_t seed, uint64_t offset, void* stream) {
  SLAM_CHECK_ARG(X && A && U, "slam_lora_a_fwd: null pointer");
  SLAM_CHECK_ARG(M > 0 && R > 0 && R <= 64 && R % 4 == 0 && K > 0 && K % 64 == 0, "slam_lora_a_fwd: need R %% 4 == 0, R <= 64, K %% 64 == 0");
  SLAM_CHECK_ARG(ldx % 8 == 0 && lda % 8 == 0 && ldu % 4 == 0 && ((uintptr_t)U % 8) == 0, "slam_lora_a_fwd: misaligned operands");
  SLAM_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && offset % 8 == 0, "slam_lora_a_fwd: bad dropout arguments");
  const unsigned th = drop_p > 0.f ? slam_drop_thresh16(drop_p) : 0u;
  const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  dim3 grid((unsigned)cdiv64(M, 32));
  hipStream_t s = (hipStream_t)stream;
  const int nt = (int)cdiv64(R, 16);
#define SLAM_LAUNCH_LA(NT_) hipLaunchKernelGGL(lora_a_fwd_kernel<NT_>, grid, dim3(256), 0, s, (const bf16_t*)X, ldx, (const bf16_t*)A, lda, \
                                              (bf16_t*)U, ldu, (int)M, (int)R, (int)K, th, inv_keep, (unsigned long long)seed, (unsigned long long)offset)
  if (nt == 1) SLAM_LAUNCH_LA(1);
  else if (nt == 2) SLAM_LAUNCH_LA(2);
  else if (nt == 3) SLAM_LAUNCH_LA(3);
  else SLAM_LAUNCH_LA(4);
#undef SLAM_LAUNCH_LA
  SLAM_CHECK_LAUNCH("slam_lora_a_fwd");
  return 0;
}

extern "C" int64_t slam_skinny_gram_workspace_bytes(int64_t M, int64_t R, int64_t C) {
  const int64_t nsplit = (M + 255) / 256;
  return nsplit * R * C * (int64_t)sizeof(float);
}

extern "C" int slam_skinny_gram(const void* S, int64_t lds_, const void* X, int64_t ldx, float* out,
                                int64_t out_ld_r, int64_t out_ld_c, int64_t M, int64_t R, int64_t C, float alpha,
                                int accumulate, float drop_p, uint64_t seed, uint64_t offset, float* workspace,
                                void* stream) {
  SLAM_CHECK_ARG(S && X && out && workspace, "slam_skinny_gram: null pointer");
  SLAM_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && offset % 8 == 0, "slam_skinny_gram: bad dropout arguments");
  const unsigned th = drop_p > 0.f ? slam_drop_thresh16(drop_p) : 0u;
  if (drop_p > 0.f) alpha *= 1.0f / (1.0f - drop_p);
  SLAM_CHECK_ARG(R == 8 || R == 16 || R == 32 || R == 64, "slam_skinny_gram: R=%ld must be 8, 16, 32 or 64", (long)R);
  SLAM_CHECK_ARG(M > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0, "slam_skinny_gram: C and ldx must be multiples of 8");
  SLAM_CHECK_ARG(lds_ % (R / 4) == 0 && ((uintptr_t)S % (R / 2)) == 0, "slam_skinny_gram: S must be aligned to R/4 elements (vector loads)");
  SLAM_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)workspace % 16) == 0, "slam_skinny_gram: X/workspace must be 16-byte aligned");
  const int rows_per_split = 256;
  const int nsplit = (int)((M + rows_per_split - 1) / rows_per_split);
  dim3 grid((unsigned)cdiv64(C, 512), (unsigned)nsplit);
  hipStream_t s = (hipStream_t)stream;
  switch (R) {
    case 8: hipLaunchKernelGGL(skinny_gram_kernel<2>, grid, dim3(256), 0, s, (const bf16_t*)S, lds_, (const bf16_t*)X, ldx, workspace, (int)M, (int)R, (int)C, rows_per_split, th, (unsigned long long)seed, (unsigned long long)offset); break;
    case 16: hipLaunchKernelGGL(skinny_gram_kernel<4>, grid, dim3(256), 0, s, (const bf16_t*)S, lds_, (const bf16_t*)X, ldx, workspace, (int)M, (int)R, (int)C, rows_per_split, th, (unsigned long long)seed, (unsigned long long)offset); break;
    case 32: hipLaunchKernelGGL(skinny_gram_kernel<8>, grid, dim3(256), 0, s, (const bf16_t*)S, lds_, (const bf16_t*)X, ldx, workspace, (int)M, (int)R, (int)C, rows_per_split, th, (unsigned long long)seed, (unsigned long long)offset); break;
    default: hipLaunchKernelGGL(skinny_gram_kernel<16>, grid, dim3(256), 0, s, (const bf16_t*)S, lds_, (const bf16_t*)X, ldx, workspace, (int)M, (int)R, (int)C, rows_per_split, th, (unsigned long long)seed, (unsigned long long)offset); break;
  }
  int64_t g = cdiv64(R * C, 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(skinny_gram_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, workspace, out, out_ld_r, out_ld_c,
                     (int)R, (int)C, nsplit, alpha, accumulate);
  SLAM_CHECK_LAUNCH("slam_skinny_gram");
  return 0;
}
