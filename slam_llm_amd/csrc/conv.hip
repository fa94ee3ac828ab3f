// Grouped positional convolution of the HuBERT / WavLM encoders as ONE implicit-GEMM launch (gfx950).
//
// Reference: fairseq's `pos_conv` (nn.Conv1d(d, d, kernel_size = 128, padding = 64, groups = 16) + SamePad + GELU) followed by the
// residual add -- src/slam_llm/models/wavlm/WavLM.py:378-386 (make_conv_pos), :575-580 (x_conv = self.pos_conv(x.transpose(1, 2));
// x = x + x_conv), the same graph in fairseq's HuBERT reached through src/slam_llm/models/slam_model.py:335-341:
//     x[b, t, g*C + co] = h[b, t, g*C + co] + gelu(bias[g*C + co] + sum_{j < K} sum_{ci < C} W[g*C + co, ci, j] * h[b, t + j - K/2, g*C + ci])
// with C = d / groups channels per group and zero padding outside [0, T).
//
// Round 2 ran this as 16 x (im2col into a [B*T, 128*C] buffer + one N = C GEMM): 16 x 147 MB of column buffers and a 64-column
// output tile that cannot feed the matrix pipe (1.57 ms at 96 TFLOP/s for 6 x 1499 frames, profiles/r02_c4_hubert.md).  Here a
// workgroup owns 256 time steps x the C output channels of one (batch, group):
//   * the input window (256 + K - 1 rows x C channels, zero filled outside the clip) is loaded into LDS ONCE: tap j of the
//     convolution is the same window shifted by j rows, so the im2col matrix exists only as LDS addressing (implicit GEMM);
//   * the weights are pre-packed tap-major ([group][tap][co][ci padded to a multiple of 32], 8 KiB per tap at C = 64) and streamed
//     from L2 straight into the MFMA operand registers, one tap ahead of the products (register double buffering);
//   * 4 waves x (64 time steps x C channels) accumulate with v_mfma_f32_16x16x32_bf16 (operands swapped so that a lane ends
//     up with 4 consecutive output channels), K = 128 taps x C channels deep;
//   * epilogue: + bias, exact-erf GELU, + residual taken from the window already in LDS, 8-byte bf16 stores.
// Workgroups are numbered XCD-aware (consecutive ids = time tiles of one (batch, group), groups slowest) so that the 1 MiB of one
// group's weights is shared out of one XCD's L2.
#include "common.h"

namespace {

// Arguments beyond the forward's: `pad` = rows of left padding (taps / 2 for the forward; taps - 1 - taps / 2 for the ADJOINT, which is
// the same kernel run on dL/d(conv output) with tap-reversed, channel-transposed weights: dL/dh[s] = sum_j W_j^T d[s - j + taps/2]);
// `bias` nullable; `act` 1 = GELU, 0 = none; `res` nullable: the residual comes from this tensor instead of the window (the adjoint adds
// the skip path's gradient); `pre` nullable: also write the pre-activation conv + bias (the training forward keeps it for the backward).
template <int C>   // channels per group: 32 | 48 | 64 | 80 (d = 512 / 768 / 1024 / 1280 with 16 groups)
__global__ __launch_bounds__(256, (C <= 64 ? 2 : 1)) void pos_conv_kernel(const bf16_t* __restrict__ h, int64_t ldh, const bf16_t* __restrict__ wpk,
                                                          const float* __restrict__ bias, bf16_t* __restrict__ x, int64_t ldx,
                                                          bf16_t* __restrict__ pre, int64_t ldpre, const bf16_t* __restrict__ res,
                                                          int64_t ldr, int B, int T, int K, int pad, int act, int tiles_t) {
  constexpr int KP = (C + 31) / 32 * 32;   // channels per tap as the MFMA sees them (zero padded)
  constexpr int KS = KP / 32;              // 32-deep k-steps per tap
  constexpr int NF = C / 16;               // 16-wide output-channel fragments
  constexpr int BM = 256;                  // time steps per workgroup
  constexpr int RB = KP * 2 + 16;          // LDS row pitch in bytes (padded: rows 16 apart do not share banks)
  extern __shared__ __attribute__((aligned(16))) char win[];   // (BM + K - 1) rows x RB

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fg = lane >> 4;
  // ---- XCD-aware block numbering (bijective): XCD x owns a contiguous run of logical ids ----
  int bid = blockIdx.x;
  {
    const int n = gridDim.x, xcd = bid & 7, q = n >> 3, r = n & 7;
    bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tt = bid % tiles_t;
  const int b = (bid / tiles_t) % B;
  const int g = bid / (tiles_t * B);
  const int t0 = tt * BM;
  const int rows = BM + K - 1;

  // ---- window: row r holds h[b, t0 - pad + r, g*C .. g*C + C) (zeros outside the clip and in the channel padding) ----
  constexpr int CH = RB / 16;              // 16-byte chunks per LDS row (incl. padding)
  constexpr int CHV = C * 2 / 16;          // chunks that carry channels
  for (int i = tid; i < rows * CH; i += 256) {
    const int r = i / CH, c = i % CH;
    const int t = t0 - pad + r;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (c < CHV && t >= 0 && t < T) v = *reinterpret_cast<const uint4*>(h + ((int64_t)b * T + t) * ldh + g * C + c * 8);
    *reinterpret_cast<uint4*>(win + r * RB + c * 16) = v;
  }
  __syncthreads();

  f32x4_t acc[4][NF];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int n = 0; n < NF; n++) acc[i][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // weights of tap j: [C rows (co)][KP] bf16; this lane's fragment (n, ks) = row n*16 + frow, channels (ks*4 + fg)*8 .. +7
  const bf16_t* wg = wpk + (int64_t)g * K * C * KP + (int64_t)frow * KP + fg * 8;
  auto load_w = [&](int j, bf16x8_t (&w)[NF][KS]) {
    const bf16_t* wj = wg + (int64_t)j * C * KP;
#pragma unroll
    for (int n = 0; n < NF; n++)
#pragma unroll
      for (int ks = 0; ks < KS; ks++) w[n][ks] = *reinterpret_cast<const bf16x8_t*>(wj + n * 16 * KP + ks * 32);
  };
  const char* arow = win + (wave * 64 + frow) * RB + fg * 16;   // + (i*16 + j) * RB + ks * 64

  bf16x8_t w0[NF][KS], w1[NF][KS];
  load_w(0, w0);
  auto tap = [&](int j, bf16x8_t (&wc)[NF][KS], bf16x8_t (&wn)[NF][KS]) {
    load_w(min(j + 1, K - 1), wn);   // one tap ahead (past the end: the last tap again, never used)
    bf16x8_t a[4][KS];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int ks = 0; ks < KS; ks++) a[i][ks] = *reinterpret_cast<const bf16x8_t*>(arow + (i * 16 + j) * RB + ks * 64);
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int n = 0; n < NF; n++) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[n][ks], a[i][ks], acc[i][n], 0, 0, 0);
  };
  int j = 0;
  for (; j + 1 < K; j += 2) {
    tap(j, w0, w1);
    tap(j + 1, w1, w0);
  }
  if (j < K) tap(j, w0, w1);

  // ---- epilogue: lane holds time step (wave*64 + i*16 + frow), output channels n*16 + fg*4 .. +3 ----
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int tl = wave * 64 + i * 16 + frow;
    const int t = t0 + tl;
    if (t >= T) continue;
    // residual: the window row of time step t (= h[b, t, g*C ..], forward) or the caller's tensor (adjoint: the skip gradient)
    const bf16_t* rrow = res ? res + ((int64_t)b * T + t) * ldr + g * C : reinterpret_cast<const bf16_t*>(win + (tl + pad) * RB);
    bf16_t* xrow = x + ((int64_t)b * T + t) * ldx + g * C;
#pragma unroll
    for (int n = 0; n < NF; n++) {
      const int co = n * 16 + fg * 4;
      const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + g * C + co) : make_float4(0.f, 0.f, 0.f, 0.f);
      const u16x4_t r4 = *reinterpret_cast<const u16x4_t*>(rrow + co);
      float v[4] = {acc[i][n][0] + bv.x, acc[i][n][1] + bv.y, acc[i][n][2] + bv.z, acc[i][n][3] + bv.w};
      if (pre) {
        uint2 pz;
        pz.x = pack2bf(v[0], v[1]);
        pz.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(pre + ((int64_t)b * T + t) * ldpre + g * C + co) = pz;
      }
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = (act ? gelu_erf(v[e]) : v[e]) + bf2f(r4[e]);
      uint2 o;
      o.x = pack2bf(v[0], v[1]);
      o.y = pack2bf(v[2], v[3]);
      *reinterpret_cast<uint2*>(xrow + co) = o;
    }
  }
}

template <int C>
int launch_pos_conv(const bf16_t* h, int64_t ldh, const bf16_t* wpk, const float* bias, bf16_t* x, int64_t ldx, bf16_t* pre, int64_t ldpre,
                    const bf16_t* res, int64_t ldr, int B, int T, int G, int K, int pad, int act, hipStream_t s) {
  constexpr int KP = (C + 31) / 32 * 32;
  const int lds = (256 + K - 1) * (KP * 2 + 16);
  auto kern = pos_conv_kernel<C>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      slam_set_error("slam_pos_conv_fwd: cannot raise the LDS limit to %d", lds);
      return -2;
    }
  }
  const int tiles_t = (T + 255) / 256;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_t * B * G)), dim3(256), lds, s, h, ldh, wpk, bias, x, ldx, pre, ldpre, res, ldr, B, T, K, pad,
                     act, tiles_t);
  SLAM_CHECK_LAUNCH("slam_pos_conv_fwd");
  return 0;
}

}  // namespace

extern "C" int slam_pos_conv_supported(int64_t channels_per_group, int64_t taps) {
  return (channels_per_group == 32 || channels_per_group == 48 || channels_per_group == 64 || channels_per_group == 80) && taps >= 1 && taps <= 256 ? 1 : 0;
}

extern "C" int slam_pos_conv_fwd(const void* h, int64_t ldh, const void* w_packed, const float* bias, void* x, int64_t ldx, void* pre,
                                 int64_t ldpre, const void* residual, int64_t ldr, int64_t B, int64_t T, int64_t groups,
                                 int64_t channels_per_group, int64_t taps, int64_t pad, int act, void* stream) {
  SLAM_CHECK_ARG(h && w_packed && x, "slam_pos_conv_fwd: null pointer");
  SLAM_CHECK_ARG(B > 0 && T > 0 && groups > 0 && B * T < (1ll << 31) && groups * B * ((T + 255) / 256) < (1ll << 31), "slam_pos_conv_fwd: bad shape");
  SLAM_CHECK_ARG(slam_pos_conv_supported(channels_per_group, taps) == 1,
                 "slam_pos_conv_fwd: %ld channels per group / %ld taps unsupported (32 | 48 | 64 | 80 channels, <= 256 taps)", (long)channels_per_group, (long)taps);
  SLAM_CHECK_ARG(pad >= 0 && pad < taps && (act == 0 || act == 1), "slam_pos_conv_fwd: pad %ld outside [0, taps) or act %d not 0 | 1", (long)pad, act);
  const int64_t d = groups * channels_per_group;
  SLAM_CHECK_ARG(ldh % 8 == 0 && ldx % 4 == 0 && ldh >= d && ldx >= d && ((uintptr_t)h % 16) == 0 && ((uintptr_t)x % 8) == 0 &&
                     ((uintptr_t)w_packed % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0),
                 "slam_pos_conv_fwd: leading dimensions / alignment");
  SLAM_CHECK_ARG(!pre || (ldpre % 4 == 0 && ldpre >= d && ((uintptr_t)pre % 8) == 0), "slam_pos_conv_fwd: bad pre-activation buffer");
  SLAM_CHECK_ARG(!residual || (ldr % 4 == 0 && ldr >= d && ((uintptr_t)residual % 8) == 0), "slam_pos_conv_fwd: bad residual buffer");
  SLAM_CHECK_ARG(h != x, "slam_pos_conv_fwd: in-place operation is not supported (neighbouring tiles read each other's rows)");
  const bf16_t* hp = (const bf16_t*)h;
  const bf16_t* wp = (const bf16_t*)w_packed;
  const bf16_t* rp = (const bf16_t*)residual;
  bf16_t* xp = (bf16_t*)x;
  bf16_t* pp = (bf16_t*)pre;
  hipStream_t s = (hipStream_t)stream;
  switch ((int)channels_per_group) {
    case 32: return launch_pos_conv<32>(hp, ldh, wp, bias, xp, ldx, pp, ldpre, rp, ldr, (int)B, (int)T, (int)groups, (int)taps, (int)pad, act, s);
    case 48: return launch_pos_conv<48>(hp, ldh, wp, bias, xp, ldx, pp, ldpre, rp, ldr, (int)B, (int)T, (int)groups, (int)taps, (int)pad, act, s);
    case 80: return launch_pos_conv<80>(hp, ldh, wp, bias, xp, ldx, pp, ldpre, rp, ldr, (int)B, (int)T, (int)groups, (int)taps, (int)pad, act, s);
    default: return launch_pos_conv<64>(hp, ldh, wp, bias, xp, ldx, pp, ldpre, rp, ldr, (int)B, (int)T, (int)groups, (int)taps, (int)pad, act, s);
  }
}
