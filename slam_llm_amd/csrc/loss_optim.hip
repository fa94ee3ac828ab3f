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
    // a row carries a label iff its target is a class index: `ignore_index` and any other negative value (which torch's
    // CrossEntropyLoss rejects with a device assert) both become -1 -- the count, the loss rows and the labelled-rows selection of
    // the model (targets >= 0) are then the same predicate
    const bool valid = l != ignore_index && l >= 0;
    if (valid) cnt++;
    tgt[r] = valid ? (int32_t)l : -1;
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

// Device-side selection of the rows that carry a label (round 6: takes the host out of the training step -- the label count used
// to be read back through pinned memory and the row list came from a torch argsort, which blocked hipGraph capture).
//   rows[i]   (i < cap)  = index of the i-th row with targets >= 0, in row order; -1 for i >= count (slam_gather_rows_bf16 turns a
//                          negative index into a zero row: padding rows are zero activations with an ignored target)
//   tsel[i]   (i < cap)  = targets[rows[i]], -1 for the padding
//   inv[r]    (r < M)    = position of row r in `rows`, -1 when row r carries no label (or did not fit below cap)
//   count[0]             = number of rows with a label (may exceed cap: the caller's bound was too small -- its problem to detect)
// One workgroup: M is a few 10^4, the scan is chunked 1024 rows at a time with a wave-level prefix + an LDS prefix over the 16 waves.
__global__ __launch_bounds__(1024) void label_rows_kernel(const int32_t* __restrict__ tgt, int M, int cap, int32_t* __restrict__ rows,
                                                          int32_t* __restrict__ tsel, int32_t* __restrict__ inv,
                                                          int32_t* __restrict__ count) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int r0 = 0; r0 < M; r0 += 1024) {
    const int r = r0 + tid;
    const int t = r < M ? tgt[r] : -1;
    const bool has = t >= 0;
    const unsigned long long ball = __ballot(has);
    const int before = __popcll(ball & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(ball);
    __syncthreads();
    int wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
      const int c = wsum[w];
      if (w < wave) wbase += c;
      tot += c;
    }
    const int base = base_s;
    const int pos = base + wbase + before;
    if (r < M) {
      const bool fits = has && pos < cap;
      inv[r] = fits ? pos : -1;
      if (fits) { rows[pos] = r; tsel[pos] = t; }
    }
    __syncthreads();
    if (tid == 0) base_s = base + tot;
    __syncthreads();
  }
  const int n = base_s;
  for (int i = n + tid; i < cap; i += 1024) { rows[i] = -1; tsel[i] = -1; }
  if (tid == 0) *count = n;
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
                                                    float bc1, float bc2_sqrt, float gscale, const float* __restrict__ hyper) {
  if (hyper) {     // slam_adamw_step_dev: lr and the two bias corrections come from device memory (a captured step replays its arguments)
    lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2];
  }
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

// AnyPrecisionAdamW (src/slam_llm/policies/anyprecision_optimizer.py:73-178) with bf16 momentum / variance (/ Kahan
// compensation) as pipeline/finetune.py:237-245 builds it.  The reference is a chain of tensor ops, each rounding to ITS
// tensor's dtype; the same roundings are applied here in registers, in the same order:
//   p.mul_(1 - lr*wd);  m.mul_(b1).add_(g, alpha=1-b1);  v.mul_(b2).addcmul_(g, g, value=1-b2);
//   denom = (v.sqrt() / dc).add_(eps);  p.addcdiv_(m, denom, value=-step_size)        [or the Kahan form :152-160]
// PBF = parameters are bf16 in the reference (its pure_bf16 route): the fp32 master buffer then only ever holds
// bf16-representable values and the gradient is rounded to bf16 first (p.grad has the parameter's dtype there).
__device__ __forceinline__ float rbf(float x) { return bf2f(f2bf(x)); }

template <bool PBF, bool KAHAN>
__global__ __launch_bounds__(256) void adamw_anyprecision_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                 bf16_t* __restrict__ m, bf16_t* __restrict__ v,
                                                                 bf16_t* __restrict__ comp, bf16_t* __restrict__ p_bf16,
                                                                 int64_t n, float decay, int use_decay, float b1, float a1,
                                                                 float b2, float a2, float dc, float eps, float neg_step) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = PBF ? rbf(g[i]) : g[i];
    float pi = p[i];
    if (use_decay) {
      pi = pi * decay;
      if (PBF) pi = rbf(pi);
    }
    const float mi = rbf(rbf(bf2f(m[i]) * b1) + a1 * gi);
    const float vi = rbf(rbf(bf2f(v[i]) * b2) + (a2 * gi) * gi);
    const float den = rbf(rbf(rbf(sqrtf(vi)) / dc) + eps);
    if (KAHAN) {
      float ci = rbf(bf2f(comp[i]) + neg_step * (mi / den));
      const float tmp = pi;
      pi = pi + ci;
      if (PBF) pi = rbf(pi);
      float dlt = tmp - pi;
      if (PBF) dlt = rbf(dlt);
      ci = rbf(ci + dlt);
      comp[i] = f2bf(ci);
    } else {
      pi = pi + neg_step * (mi / den);
      if (PBF) pi = rbf(pi);
    }
    p[i] = pi;
    m[i] = f2bf(mi);
    v[i] = f2bf(vi);
    if (p_bf16) p_bf16[i] = f2bf(pi);
  }
}

}  // namespace

extern "C" int slam_adamw_anyprecision_step(float* param, const float* grad, void* exp_avg_bf16, void* exp_avg_sq_bf16,
                                            void* compensation_bf16, void* param_bf16, int64_t n, float decay,
                                            int use_decay, float beta1, float one_minus_beta1, float beta2,
                                            float one_minus_beta2, float denom_correction, float eps, float neg_step_size,
                                            int params_are_bf16, void* stream) {
  SLAM_CHECK_ARG(param && grad && exp_avg_bf16 && exp_avg_sq_bf16 && n > 0, "slam_adamw_anyprecision_step: bad arguments");
  SLAM_CHECK_ARG(denom_correction > 0.f, "slam_adamw_anyprecision_step: denom_correction must be > 0 (step >= 1)");
  int64_t g = cdiv64(n, 256);
  if (g > 8192) g = 8192;
  hipStream_t s = (hipStream_t)stream;
  bf16_t *m = (bf16_t*)exp_avg_bf16, *v = (bf16_t*)exp_avg_sq_bf16, *c = (bf16_t*)compensation_bf16, *pb = (bf16_t*)param_bf16;
#define SLAM_ANYP(PBF, KAHAN)                                                                                              \
  hipLaunchKernelGGL((adamw_anyprecision_kernel<PBF, KAHAN>), dim3((unsigned)g), dim3(256), 0, s, param, grad, m, v, c, pb, n, \
                     decay, use_decay, beta1, one_minus_beta1, beta2, one_minus_beta2, denom_correction, eps, neg_step_size)
  if (params_are_bf16) {
    if (c) SLAM_ANYP(true, true); else SLAM_ANYP(true, false);
  } else {
    if (c) SLAM_ANYP(false, true); else SLAM_ANYP(false, false);
  }
#undef SLAM_ANYP
  SLAM_CHECK_LAUNCH("slam_adamw_anyprecision_step");
  return 0;
}

extern "C" int slam_ce_targets(const int64_t* labels, int32_t* targets, int32_t* n_valid, int64_t B,
                               int64_t T, int64_t ignore_index, void* stream) {
  SLAM_CHECK_ARG(labels && targets && n_valid && B > 0 && T > 0, "slam_ce_targets: bad arguments");
  hipLaunchKernelGGL(ce_targets_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, labels, targets,
                     n_valid, B * T, (int)T, (int)ignore_index);
  SLAM_CHECK_LAUNCH("slam_ce_targets");
  return 0;
}

extern "C" int slam_label_rows(const int32_t* targets, int64_t M, int64_t cap, int32_t* rows, int32_t* tsel, int32_t* inv, int32_t* count,
                               void* stream) {
  SLAM_CHECK_ARG(targets && rows && tsel && inv && count, "slam_label_rows: null pointer");
  SLAM_CHECK_ARG(M > 0 && M < (1ll << 31) && cap > 0 && cap <= M, "slam_label_rows: need 0 < cap <= M < 2^31 (M=%ld cap=%ld)", (long)M, (long)cap);
  hipLaunchKernelGGL(label_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, targets, (int)M, (int)cap, rows, tsel, inv, count);
  SLAM_CHECK_LAUNCH("slam_label_rows");
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
                     sqrtf(bc2), grad_scale, (const float*)nullptr);
  SLAM_CHECK_LAUNCH("slam_adamw_step");
  return 0;
}

// host-side helper: the three words slam_adamw_step_dev reads, formed exactly like slam_adamw_step forms them (float powf / sqrtf), so that
// a captured step is bit-identical to the eager one; `out3` is HOST memory (the caller copies it to the device buffer)
extern "C" int slam_adamw_hyper(float lr, float beta1, float beta2, int64_t step, float* out3) {
  SLAM_CHECK_ARG(out3 && step >= 1, "slam_adamw_hyper: need a host buffer of 3 floats and step >= 1");
  out3[0] = lr;
  out3[1] = 1.0f - powf(beta1, (float)step);
  out3[2] = sqrtf(1.0f - powf(beta2, (float)step));
  return 0;
}

// the same kernel with lr, 1 - beta1^step and sqrt(1 - beta2^step) read from device memory (hyper[0..2], fp32): what a step captured in a
// hipGraph needs -- the host (LambdaLR, the step counter) writes the three words before each replay
extern "C" int slam_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16, int64_t n,
                                   const float* hyper, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                   void* stream) {
  SLAM_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && hyper && n > 0, "slam_adamw_step_dev: bad arguments");
  int64_t g = cdiv64(n, 256);
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                     (bf16_t*)param_bf16, n, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, grad_scale, hyper);
  SLAM_CHECK_LAUNCH("slam_adamw_step_dev");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Tall-skinny gram product for the LoRA gradients (peft Linear backward: dA = (dy.sB)^T dropout(x), dB = s dy^T (xA^T)):
//   out[r, c] (+)= alpha * sum_m S[m, r] * X'[m, c]      S: [M, R] bf16, R % 8 == 0, R <= 64;  X: [M, C] bf16
// The reduction runs over the token dimension M (~12 k): both MFMA operands would need M-contiguous (transposed)
// layouts.  Instead of writing transposed copies to HBM (the first version of this path) or doing the product on the
// VALU (1.5 G fp32 FMA for one dA: 160 us), 32-row tiles of X and S are staged ROW-MAJOR in LDS (coalesced 16-byte
// global loads, lora_dropout mask recomputed once per element on the way) and the MFMA fragments are gathered with
// 2-byte LDS reads; row strides are padded so that the 4 row groups of a fragment read land in disjoint banks.
// X is read from HBM exactly once.  M is split over gridDim.y; partials are reduced in a fixed order (bit-reproducible).
// ------------------------------------------------------------------------------------------------------------
namespace {

constexpr int GM_CB = 256;            // columns of X per workgroup (4 waves x 4 MFMA column tiles)
constexpr int GM_XLD = GM_CB + 16;    // 544-byte rows: consecutive rows shift by 8 banks -> conflict-free gathers

template <int NT>
__global__ __launch_bounds__(256) void gram_mfma_kernel(const bf16_t* __restrict__ S, int64_t lds_,
                                                        const bf16_t* __restrict__ X, int64_t ldx,
                                                        float* __restrict__ ws, int M, int R, int C, int rows_per_split,
                                                        unsigned thresh16, unsigned long long seed,
                                                        unsigned long long offset, const unsigned long long* salt) {
  seed = slam_salted(seed, salt);
  constexpr int SLD = NT == 1 ? 16 : (NT == 2 ? 48 : 80);   // dword strides 8 / 24 / 40: 4 consecutive rows -> disjoint banks
  constexpr int SCH = NT * 2;                               // 16-byte chunks per S row
  __shared__ __attribute__((aligned(16))) bf16_t xs[32 * GM_XLD];
  __shared__ __attribute__((aligned(16))) bf16_t ss[32 * SLD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int c_blk = blockIdx.x * GM_CB;
  const int m0 = blockIdx.y * rows_per_split, m1 = min(M, m0 + rows_per_split);
  f32x4_t acc[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int t = 0; t < 4; t++) acc[nt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  u16x8_t xreg[4], sreg;
  auto gload = [&](int mb) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int id = tid + i * 256, row = id >> 5, cc = id & 31;
      const int m = mb + row, c = c_blk + cc * 8;
      u16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (m < m1 && c < C) {
        v = *reinterpret_cast<const u16x8_t*>(X + (int64_t)m * ldx + c);
        if (thresh16) {
          const unsigned keep = slam_keep8(seed, offset + (unsigned long long)m * (unsigned long long)C + c, thresh16);
#pragma unroll
          for (int e = 0; e < 8; e++)
            if (!((keep >> e) & 1u)) v[e] = 0;
        }
      }
      xreg[i] = v;
    }
    if (tid < 32 * SCH) {
      const int row = tid / SCH, jc = tid % SCH;
      const int m = mb + row;
      u16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (m < m1 && jc * 8 < R) v = *reinterpret_cast<const u16x8_t*>(S + (int64_t)m * lds_ + jc * 8);
      sreg = v;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int id = tid + i * 256, row = id >> 5, cc = id & 31;
      *reinterpret_cast<u16x8_t*>(&xs[row * GM_XLD + cc * 8]) = xreg[i];
    }
    if (tid < 32 * SCH) {
      const int row = tid / SCH, jc = tid % SCH;
      *reinterpret_cast<u16x8_t*>(&ss[row * SLD + jc * 8]) = sreg;
    }
  };

  if (m0 < m1) gload(m0);
  for (int mb = m0; mb < m1; mb += 32) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (mb + 32 < m1) gload(mb + 32);
    // MFMA k-slot (g, e) <-> tile row e*4 + g for BOTH operands (any bijection works; this one is bank-friendly)
    u16x8_t af[NT], bfr[4];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
      for (int e = 0; e < 8; e++) af[nt][e] = ss[(e * 4 + g) * SLD + nt * 16 + li];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int e = 0; e < 8; e++) bfr[t][e] = xs[(e * 4 + g) * GM_XLD + (wave * 4 + t) * 16 + li];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
      for (int t = 0; t < 4; t++)
        acc[nt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[nt]),
                                                            __builtin_bit_cast(bf16x8_t, bfr[t]), acc[nt][t], 0, 0, 0);
  }
  // acc[nt][t][i] = partial out[j = nt*16 + 4g + i][c = c_blk + (wave*4 + t)*16 + li]
  float* w = ws + (int64_t)blockIdx.y * R * C;
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int c = c_blk + (wave * 4 + t) * 16 + li;
      if (c >= C) continue;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int j = nt * 16 + 4 * g + i;
        if (j < R) w[(int64_t)j * C + c] = acc[nt][t][i];
      }
    }
}

void gram_plan(int64_t M, int64_t C, int* rows_per_split, int* nsplit) {
  const int64_t ncb = cdiv64(C, GM_CB);
  int64_t ns = cdiv64(512, ncb);
  if (ns > cdiv64(M, 32)) ns = cdiv64(M, 32);
  if (ns < 1) ns = 1;
  const int64_t rps = cdiv64(cdiv64(M, ns), 32) * 32;
  *rows_per_split = (int)rps;
  *nsplit = (int)cdiv64(M, rps);
}

__global__ __launch_bounds__(256) void skinny_gram_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                                 int64_t ld_r, int64_t ld_c, int R, int C, int nsplit,
                                                                 float alpha, int accumulate) {
  const int64_t total = (int64_t)R * C;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float s = 0.f;
#pragma unroll 8   // the partial loads are independent: keep 8 in flight (the adds stay in fixed order)
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
__global__ __launch_bounds__((NT <= 2 ? 8 : 4) * 64, 2) void lora_a_fwd_kernel(const bf16_t* __restrict__ X, int64_t ldx,
                                                         const bf16_t* __restrict__ A, int64_t lda,
                                                         bf16_t* __restrict__ U, int64_t ldu, int M, int R, int Rpad, int K,
                                                         unsigned thresh16, float inv_keep, unsigned long long seed,
                                                         unsigned long long offset, const unsigned long long* salt) {
  seed = slam_salted(seed, salt);
  constexpr int NWV = NT <= 2 ? 8 : 4;   // waves splitting K (8 when the LDS reduction buffer allows: more loads in flight)
  __shared__ float red[NWV][2 * NT][256 + 4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * 32;
  const int kw = (int)(((K / 64 + NWV - 1) / NWV) * 64);
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
  // x is streamed in batches of LA_UNR k-steps: all 4 * LA_UNR 16-byte loads of a batch are issued before the first product waits
  // (the kernel is latency-bound on HBM: 4 loads in flight per lane left it at 2.1 TB/s); A comes from L2
  constexpr int LA_UNR = NT <= 2 ? 3 : 2;
  for (int kb = k_begin; kb < k_end; kb += 64 * LA_UNR) {
    u16x8_t xf[LA_UNR][2][2];
#pragma unroll
    for (int it = 0; it < LA_UNR; it++) {
      const int k0 = min(kb + it * 64, k_end - 64);      // (a short last batch re-reads its last step; skipped below)
#pragma unroll
      for (int mt = 0; mt < 2; mt++) {
        xf[it][mt][0] = *reinterpret_cast<const u16x8_t*>(xp[mt] + k0);
        xf[it][mt][1] = *reinterpret_cast<const u16x8_t*>(xp[mt] + k0 + 8);
      }
    }
#pragma unroll
    for (int it = 0; it < LA_UNR; it++) {
      const int k0 = kb + it * 64;
      if (k0 >= k_end) break;
      u16x8_t af[NT][2];
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
              if (!((keep >> e) & 1u)) xf[it][mt][hf][e] = 0;
          }
      }
#pragma unroll
      for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[nt][0]),
                                                               __builtin_bit_cast(bf16x8_t, xf[it][mt][0]), acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[nt][1]),
                                                               __builtin_bit_cast(bf16x8_t, xf[it][mt][1]), acc[mt][nt], 0, 0, 0);
        }
    }
  }
  // acc[mt][nt][i] = u[m0 + mt*16 + r][nt*16 + 4g + i]
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) *reinterpret_cast<f32x4_t*>(&red[wave][mt * NT + nt][lane * 4]) = acc[mt][nt];
  __syncthreads();
  for (int idx = tid; idx < 32 * NT * 4; idx += NWV * 64) {   // (row, 4 consecutive ranks)
    const int row = idx / (NT * 4), jq = idx % (NT * 4);
    const int mt = row >> 4, rr = row & 15, nt = jq >> 2, gg = jq & 3;
    const int m = m0 + row, j = nt * 16 + gg * 4;
    if (m >= M || j >= R) continue;
    const int src = (gg * 16 + rr) * 4;
    f32x4_t v = *reinterpret_cast<const f32x4_t*>(&red[0][mt * NT + nt][src]);
#pragma unroll
    for (int w = 1; w < NWV; w++) {
      const f32x4_t t = *reinterpret_cast<const f32x4_t*>(&red[w][mt * NT + nt][src]);
#pragma unroll
      for (int i = 0; i < 4; i++) v[i] += t[i];
    }
    uint2 o;
    o.x = pack2bf(v[0] * inv_keep, v[1] * inv_keep);
    o.y = pack2bf(v[2] * inv_keep, v[3] * inv_keep);
    *reinterpret_cast<uint2*>(U + (int64_t)m * ldu + j) = o;
  }
  // columns [R, Rpad) of the K-extension are padding (zero rows of the extended weight meet them): zeroed here, by the kernel that
  // writes their neighbours, instead of by a strided fill launch per layer (round 5: 32 launches x 16 us per C3 step)
  const int pq = (Rpad - R) >> 2;
  for (int idx = tid; idx < 32 * pq; idx += NWV * 64) {
    const int m = m0 + idx / pq, j = R + (idx % pq) * 4;
    if (m < M) *reinterpret_cast<uint2*>(U + (int64_t)m * ldu + j) = make_uint2(0u, 0u);
  }
}


// ------------------------------------------------------------------------------------------------------------
// LoRA backward, second hop with lora_dropout live (round 4): dx[M, K] += mask . (du[M, R] . A[R, K]) / (1 - p), the mask of
// slam_dropout_bf16 at index offset + m * K + k.  Was two launches -- the rank-R product into a separate [M, K] buffer, then
// slam_dropout_bf16(accumulate) over it (96 + 290 MB of traffic per Llama layer at the C3 shape for a 193 MB update) -- now one pass over
// dx: the product on the MFMA (R / 32 k-steps, operands straight from global memory: A^T fragments stay in registers, du is L2-resident),
// fp32 values exchanged between lane rows (v_permlane16_swap, v_permlane32_swap) so that a lane owns SIXTEEN consecutive columns = two
// 16-byte loads + stores of dx and two slam_keep8 mask words.  Measured 48 us against 71 us for the two launches at [11780, 4096]
// (4.0 TB/s of algorithmic traffic).  What bounds it is instruction issue, ~1500 cycles per 16 x 64 block and wave (conversions, the
// mask's compares / selects, eight exchanges, address arithmetic): 47 k blocks on 1024 SIMDs = 33 us at 2.1 GHz.  NOT the mask hash (a
// three-multiply 32-bit hash was built, validated and measured: 51.2 -> 51.2 us, reverted) and not the width of the row segments
// (four waves on neighbouring column blocks = 512 contiguous bytes per row: 51 -> 50 us, kept); the grid is one residency round.  Rounding
// reproduces the two-launch form bit for
// bit: the product is rounded to bf16 before the mask / scale, the sum once more.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void hop_swap_rows16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

__device__ __forceinline__ void hop_swap_half32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

template <int KS>   // KS = R / 32 k-steps (R <= 64, zero-padded to a multiple of 32 by the caller's buffers)
__global__ __launch_bounds__(256) void lora_hop_drop_kernel(const bf16_t* __restrict__ DU, int64_t lddu, const bf16_t* __restrict__ AT, int64_t ldat,
                                                            bf16_t* __restrict__ DX, int64_t lddx, int M, int K, int nrb, float inv_keep,
                                                            unsigned thresh, unsigned long long seed, unsigned long long offset,
                                                            const unsigned long long* salt) {
  seed = slam_salted(seed, salt);
  // workgroup = 256 columns of dx (one 64-column block per wave, its four fragments of A^T in registers) x 16-row blocks blockIdx.y, + gridDim.y, ...
  // (a wave per 16 ROWS of one 64-column block -- 128-byte row segments at an 8 KiB pitch -- held 3.8 TB/s whatever the mask cost);
  // the next row block's du and dx pieces are requested before the current one is computed.  (The first form -- a 64 x 256 tile per
  // workgroup with all sixteen A^T fragments preloaded -- used ~200 VGPRs, ran 2 waves / SIMD and reached 2.3 TB/s.)
  typedef __attribute__((ext_vector_type(8))) __bf16 bfrag_t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int frow = lane & 15, fg = lane >> 4;
  const int n0 = (blockIdx.x * 4 + wave) * 64;     // the four waves take NEIGHBOURING 64-column blocks of the same 16 rows: 512 contiguous bytes per row
  const int nn = n0 + fg * 16;                       // this lane's 16 consecutive columns after the two exchange levels
  bfrag_t bf[4][KS];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int n = min(n0 + j * 16 + frow, K - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ks++) bf[j][ks] = *reinterpret_cast<const bfrag_t*>(AT + (int64_t)n * ldat + ks * 32 + fg * 8);
  }
  bfrag_t af_n[2][KS];                              // two row blocks requested ahead (one: 3.7 TB/s)
  u16x8_t old_n[2][2];
  const int G = gridDim.y;
  auto request = [&](bfrag_t (&a)[KS], u16x8_t (&o)[2], int rb) {
    const int m = rb * 16 + frow;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) a[ks] = *reinterpret_cast<const bfrag_t*>(DU + (int64_t)min(m, M - 1) * lddu + ks * 32 + fg * 8);
#pragma unroll
    for (int h = 0; h < 2; h++)
      if (m < M && nn + 8 * h < K) o[h] = *reinterpret_cast<const u16x8_t*>(DX + (int64_t)m * lddx + nn + 8 * h);
  };
  auto compute = [&](const bfrag_t (&af)[KS], const u16x8_t (&old)[2], int rb) {
    const int m = rb * 16 + frow;
    f32x4_t acc[4];
#pragma unroll
    for (int hh = 0; hh < 4; hh++) {
      acc[hh] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ks++)
        acc[hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[hh][ks], af[ks], acc[hh], 0, 0, 0);   // lane: row frow, columns 4 fg .. + 3 of the fragment
    }
    // two exchange levels turn "4 columns of each of four fragments" into "16 consecutive columns of fragment fg": 16-lane rows first
    // (odd rows of fragment 2 i <-> even rows of 2 i + 1: columns (fg >> 1) * 8 .. + 7 of fragment 2 i + (fg & 1)), then the 32-lane halves
    // of the two pairs.  The four lanes of a matrix row then cover 128 contiguous bytes of dx.
    float v[4][4];
#pragma unroll
    for (int hh = 0; hh < 4; hh++)
#pragma unroll
      for (int e = 0; e < 4; e++) v[hh][e] = acc[hh][e];
#pragma unroll
    for (int e = 0; e < 4; e++) { hop_swap_rows16(v[0][e], v[1][e]); hop_swap_rows16(v[2][e], v[3][e]); }
#pragma unroll
    for (int e = 0; e < 4; e++) { hop_swap_half32(v[0][e], v[2][e]); hop_swap_half32(v[1][e], v[3][e]); }
    // v[0]: columns 0-3, v[1]: 4-7, v[2]: 8-11, v[3]: 12-15 of fragment fg
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (m < M && nn + 8 * h < K) {                // K % 8 == 0 (launcher)
        const unsigned keep8 = slam_keep8(seed, offset + (unsigned long long)m * (unsigned long long)K + (unsigned long long)(nn + 8 * h), thresh);
        u16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float hop = bf2f(f2bf(v[2 * h + (e >> 2)][e & 3]));        // the [M, K] bf16 buffer of the two-launch form
          const float val = ((keep8 >> e) & 1u) ? hop * inv_keep : 0.f;
          o[e] = f2bf(bf2f(old[h][e]) + val);
        }
        *reinterpret_cast<u16x8_t*>(DX + (int64_t)m * lddx + nn + 8 * h) = o;
      }
    }
  };
  int rb = blockIdx.y;
  if (rb < nrb) request(af_n[0], old_n[0], rb);
  if (rb + G < nrb) request(af_n[1], old_n[1], rb + G);
  for (; rb < nrb; rb += 2 * G) {
    bfrag_t af[KS];
    u16x8_t old[2];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) af[ks] = af_n[0][ks];
    old[0] = old_n[0][0]; old[1] = old_n[0][1];
    if (rb + 2 * G < nrb) request(af_n[0], old_n[0], rb + 2 * G);
    compute(af, old, rb);
    if (rb + G >= nrb) break;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) af[ks] = af_n[1][ks];
    old[0] = old_n[1][0]; old[1] = old_n[1][1];
    if (rb + 3 * G < nrb) request(af_n[1], old_n[1], rb + 3 * G);
    compute(af, old, rb + G);
  }
}

}  // namespace

extern "C" int slam_lora_a_fwd(const void* X, int64_t ldx, const void* A, int64_t lda, void* U, int64_t ldu, int64_t M,
                               int64_t R, int64_t Rpad, int64_t K, float drop_p, uint64_t seed, uint64_t offset, void* stream) {
  SLAM_CHECK_ARG(X && A && U, "slam_lora_a_fwd: null pointer");
  SLAM_CHECK_ARG(M > 0 && R > 0 && R <= 64 && R % 4 == 0 && K > 0 && K % 64 == 0, "slam_lora_a_fwd: need R %% 4 == 0, R <= 64, K %% 64 == 0");
  SLAM_CHECK_ARG(Rpad >= R && Rpad % 4 == 0 && Rpad <= ldu && Rpad <= 4096, "slam_lora_a_fwd: Rpad (columns of U to define: R results + zero padding) must be a multiple of 4 in [R, ldu]");
  SLAM_CHECK_ARG(ldx % 8 == 0 && lda % 8 == 0 && ldu % 4 == 0 && ((uintptr_t)U % 8) == 0, "slam_lora_a_fwd: misaligned operands");
  SLAM_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && offset % 8 == 0, "slam_lora_a_fwd: bad dropout arguments");
  const unsigned th = drop_p > 0.f ? slam_drop_thresh16(drop_p) : 0u;
  const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  dim3 grid((unsigned)cdiv64(M, 32));
  hipStream_t s = (hipStream_t)stream;
  const int nt = (int)cdiv64(R, 16);
#define SLAM_LAUNCH_LA(NT_) hipLaunchKernelGGL(lora_a_fwd_kernel<NT_>, grid, dim3((NT_ <= 2 ? 8 : 4) * 64), 0, s, (const bf16_t*)X, ldx, (const bf16_t*)A, lda, \
                                              (bf16_t*)U, ldu, (int)M, (int)R, (int)Rpad, (int)K, th, inv_keep, (unsigned long long)seed, (unsigned long long)offset, g_slam_drop_salt)
  if (nt == 1) SLAM_LAUNCH_LA(1);
  else if (nt == 2) SLAM_LAUNCH_LA(2);
  else if (nt == 3) SLAM_LAUNCH_LA(3);
  else SLAM_LAUNCH_LA(4);
#undef SLAM_LAUNCH_LA
  SLAM_CHECK_LAUNCH("slam_lora_a_fwd");
  return 0;
}

extern "C" int slam_lora_hop_dropout(const void* DU, int64_t lddu, const void* AT, int64_t ldat, void* DX, int64_t lddx, int64_t M, int64_t K,
                                     int64_t R, float drop_p, uint64_t seed, uint64_t offset, void* stream) {
  SLAM_CHECK_ARG(DU && AT && DX, "slam_lora_hop_dropout: null pointer");
  SLAM_CHECK_ARG(M > 0 && K > 0 && K % 8 == 0 && (R == 32 || R == 64), "slam_lora_hop_dropout: need K %% 8 == 0 and R (columns of du, zero padded) in {32, 64}");
  SLAM_CHECK_ARG(lddu % 8 == 0 && ldat % 8 == 0 && lddx % 8 == 0 && lddu >= R && ldat >= R && lddx >= K, "slam_lora_hop_dropout: leading dimensions");
  SLAM_CHECK_ARG(((uintptr_t)DU % 16) == 0 && ((uintptr_t)AT % 16) == 0 && ((uintptr_t)DX % 16) == 0, "slam_lora_hop_dropout: operands must be 16-byte aligned");
  SLAM_CHECK_ARG(drop_p > 0.f && drop_p < 1.f && offset % 8 == 0, "slam_lora_hop_dropout: drop_p in (0, 1), offset %% 8 == 0");
  const unsigned th = slam_drop_thresh16(drop_p);
  const float inv_keep = 1.0f / (1.0f - drop_p);
  const int nrb = (int)cdiv64(M, 16), ncb = (int)cdiv64(K, 256);
  const int chunks = (int)std::min<int64_t>(nrb, std::max<int64_t>(1, 1024 / ncb));   // 4 workgroups per CU are resident (128 VGPRs): exactly one round
  dim3 grid((unsigned)ncb, (unsigned)chunks);
  hipStream_t s = (hipStream_t)stream;
  if (R == 32)
    hipLaunchKernelGGL(lora_hop_drop_kernel<1>, grid, dim3(256), 0, s, (const bf16_t*)DU, lddu, (const bf16_t*)AT, ldat, (bf16_t*)DX, lddx, (int)M, (int)K,
                       nrb, inv_keep, th, (unsigned long long)seed, (unsigned long long)offset, g_slam_drop_salt);
  else
    hipLaunchKernelGGL(lora_hop_drop_kernel<2>, grid, dim3(256), 0, s, (const bf16_t*)DU, lddu, (const bf16_t*)AT, ldat, (bf16_t*)DX, lddx, (int)M, (int)K,
                       nrb, inv_keep, th, (unsigned long long)seed, (unsigned long long)offset, g_slam_drop_salt);
  SLAM_CHECK_LAUNCH("slam_lora_hop_dropout");
  return 0;
}

extern "C" int64_t slam_skinny_gram_workspace_bytes(int64_t M, int64_t R, int64_t C) {
  int rps, nsplit;
  gram_plan(M, C, &rps, &nsplit);
  return (int64_t)nsplit * R * C * (int64_t)sizeof(float);
}

extern "C" int slam_skinny_gram(const void* S, int64_t lds_, const void* X, int64_t ldx, float* out,
                                int64_t out_ld_r, int64_t out_ld_c, int64_t M, int64_t R, int64_t C, float alpha,
                                int accumulate, float drop_p, uint64_t seed, uint64_t offset, float* workspace,
                                void* stream) {
  SLAM_CHECK_ARG(S && X && out && workspace, "slam_skinny_gram: null pointer");
  SLAM_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && offset % 8 == 0, "slam_skinny_gram: bad dropout arguments");
  const unsigned th = drop_p > 0.f ? slam_drop_thresh16(drop_p) : 0u;
  if (drop_p > 0.f) alpha *= 1.0f / (1.0f - drop_p);
  SLAM_CHECK_ARG(R > 0 && R % 8 == 0 && R <= 64, "slam_skinny_gram: R=%ld must be a multiple of 8, <= 64", (long)R);
  SLAM_CHECK_ARG(M > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0, "slam_skinny_gram: C and ldx must be multiples of 8");
  SLAM_CHECK_ARG(lds_ % 8 == 0 && ((uintptr_t)S % 16) == 0, "slam_skinny_gram: S rows must be 16-byte aligned");
  SLAM_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)workspace % 16) == 0, "slam_skinny_gram: X/workspace must be 16-byte aligned");
  int rows_per_split, nsplit;
  gram_plan(M, C, &rows_per_split, &nsplit);
  dim3 grid((unsigned)cdiv64(C, GM_CB), (unsigned)nsplit);
  hipStream_t s = (hipStream_t)stream;
#define SLAM_LAUNCH_GM(NT_) hipLaunchKernelGGL(gram_mfma_kernel<NT_>, grid, dim3(256), 0, s, (const bf16_t*)S, lds_, (const bf16_t*)X, ldx, \
                                              workspace, (int)M, (int)R, (int)C, rows_per_split, th, (unsigned long long)seed, (unsigned long long)offset, g_slam_drop_salt)
  if (R <= 16) SLAM_LAUNCH_GM(1);
  else if (R <= 32) SLAM_LAUNCH_GM(2);
  else SLAM_LAUNCH_GM(4);
#undef SLAM_LAUNCH_GM
  int64_t g = cdiv64(R * C, 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(skinny_gram_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, workspace, out, out_ld_r, out_ld_c,
                     (int)R, (int)C, nsplit, alpha, accumulate);
  SLAM_CHECK_LAUNCH("slam_skinny_gram");
  return 0;
}
