// Decode-time kernels for `generate()` (src/slam_llm/models/slam_model.py:409-456 -> HF LlamaForCausalLM.generate
// with a KV cache).  One decode step touches every frozen weight once for a handful of rows (batch x beams), so the
// whole step is HBM-bound; nothing here is shaped for MFMA throughput, everything for streaming bytes:
//
//   * gemm_skinny16_kernel (M <= 16) / gemm_skinny_kernel (M <= 64): y = x . W^T.  W rows are streamed once with
//     full 128-B lines per row per k-step (each lane owns 32 contiguous bytes of a row); the 16 x 16 MFMA is used
//     only as a convenient dot-product engine.  K is split -- across the 16 waves of a workgroup (M <= 16, one
//     launch, LDS reduction) or across workgroups (M > 16, x staged through LDS, partials + reduce kernel) -- so
//     that thousands of waves have loads in flight whatever N is; all reductions run in a fixed order.
//   * decode_rope_append_kernel: RoPE on the new q/k (positions per row), append k/v to the generated-token cache.
//   * attn_decode_kernel: one query per row against [prompt KV of the row's batch item | generated KV reached
//     through the beam ancestor table].  Beam hypotheses SHARE the prompt KV (stored once per batch item) and the
//     generated KV is never re-ordered: a [rows, G] int table says which physical row wrote slot j of each
//     hypothesis' history (HF re-orders the whole cache with index_select every step).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__device__ __forceinline__ f32x4_t mfma16(u16x8_t a, u16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c,
                                                 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------
// skinny GEMM
// ------------------------------------------------------------------------------------------------------------
struct SkinnyParams {
  const bf16_t* A; int64_t lda;  // activations [M, K]
  const bf16_t* B; int64_t ldb;  // weights     [N, K]
  const bf16_t* B2; int64_t ldb2; int N1;  // rows >= N1 come from B2 (e.g. LoRA A stacked under W): N = N1 + N2
  int swiglu_F;                  // > 0: B = [gate(F rows); up(F rows)], output [M, F] = silu(gate) * up
  float* part;                   // [S, M, N] fp32 partials (S > 1) or null
  void* C; int64_t ldc;
  const bf16_t* res; int64_t ldr;
  int M, N, K, kslice, out_f32;
};

__device__ __forceinline__ const bf16_t* skinny_wrow(const SkinnyParams& p, int n) {
  return n < p.N1 ? p.B + (int64_t)n * p.ldb : p.B2 + (int64_t)(n - p.N1) * p.ldb2;
}
// SwiGLU on bf16-rounded gate/up exactly like swiglu_fwd_kernel (elementwise.hip) applied to a bf16 GEMM output
__device__ __forceinline__ float swiglu1(float g, float u) {
  const float gf = bf2f(f2bf(g)), uf = bf2f(f2bf(u));
  return gf / (1.0f + __expf(-gf)) * uf;
}

__device__ __forceinline__ void skinny_store(const SkinnyParams& p, int m, int n, f32x4_t v) {
  if (p.res) {
    const u16x4_t rr = *reinterpret_cast<const u16x4_t*>(p.res + (int64_t)m * p.ldr + n);
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] += bf2f(rr[i]);
  }
  if (p.out_f32) {
    *reinterpret_cast<f32x4_t*>((float*)p.C + (int64_t)m * p.ldc + n) = v;
  } else {
    uint2 o;
    o.x = pack2bf(v[0], v[1]);
    o.y = pack2bf(v[2], v[3]);
    *reinterpret_cast<uint2*>((bf16_t*)p.C + (int64_t)m * p.ldc + n) = o;
  }
}

constexpr int KC = 256;       // k elements staged per LDS chunk
constexpr int XLD = KC + 8;   // LDS row stride (elements): 528 B -> the 16 rows of a fragment read hit 16 distinct bank groups

template <int MT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(SkinnyParams p) {
  __shared__ __attribute__((aligned(16))) bf16_t xs[MT * 16 * XLD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 64 + wave * 16;
  const int k_begin = blockIdx.y * p.kslice;
  const int k_end = min(p.K, k_begin + p.kslice);
  const int NO = p.swiglu_F ? p.swiglu_F : p.N;  // output columns
  const int nrow = min(n0 + r, NO - 1);
  const bf16_t* wp = skinny_wrow(p, nrow) + g * 16;
  const bf16_t* wp2 = p.swiglu_F ? p.B + (int64_t)(p.swiglu_F + nrow) * p.ldb + g * 16 : nullptr;  // "up" rows
  f32x4_t acc[MT], acc2[MT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) acc[mt] = acc2[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int k0 = k_begin; k0 < k_end; k0 += KC) {
    const int kc = min(KC, k_end - k0);  // multiple of 64
    u16x8_t w[KC / 64][2], w2[KC / 64][2];
#pragma unroll
    for (int ks = 0; ks < KC / 64; ks++) {
      if (ks * 64 < kc) {  // lane (r, g) owns bytes [g*32, g*32+32) of row r's 128-B line of this k-step
        w[ks][0] = *reinterpret_cast<const u16x8_t*>(wp + k0 + ks * 64);
        w[ks][1] = *reinterpret_cast<const u16x8_t*>(wp + k0 + ks * 64 + 8);
        if (wp2) {
          w2[ks][0] = *reinterpret_cast<const u16x8_t*>(wp2 + k0 + ks * 64);
          w2[ks][1] = *reinterpret_cast<const u16x8_t*>(wp2 + k0 + ks * 64 + 8);
        }
      }
    }
    __syncthreads();  // the previous chunk's LDS reads are complete
    for (int idx = tid; idx < MT * 16 * (KC / 8); idx += 256) {
      const int row = idx / (KC / 8), c = idx % (KC / 8);
      u16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (row < p.M && c * 8 < kc) v = *reinterpret_cast<const u16x8_t*>(p.A + (int64_t)row * p.lda + k0 + c * 8);
      *reinterpret_cast<u16x8_t*>(&xs[row * XLD + c * 8]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KC / 64; ks++) {
      if (ks * 64 < kc) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          const bf16_t* xp = &xs[(mt * 16 + r) * XLD + ks * 64 + g * 16];
          const u16x8_t x0 = *reinterpret_cast<const u16x8_t*>(xp);
          const u16x8_t x1 = *reinterpret_cast<const u16x8_t*>(xp + 8);
          acc[mt] = mfma16(w[ks][0], x0, acc[mt]);  // D[n = 4g+i][m = r] : both operands use the same k permutation
          acc[mt] = mfma16(w[ks][1], x1, acc[mt]);
          if (wp2) {
            acc2[mt] = mfma16(w2[ks][0], x0, acc2[mt]);
            acc2[mt] = mfma16(w2[ks][1], x1, acc2[mt]);
          }
        }
      }
    }
  }
  const int n = n0 + 4 * g;
  if (n < NO) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      const int m = mt * 16 + r;
      if (m >= p.M) continue;
      if (p.part) {  // partial layout [S][M][N]; in swiglu mode N = 2F: gate columns then up columns
        float* pr = p.part + ((int64_t)blockIdx.y * p.M + m) * p.N;
        *reinterpret_cast<f32x4_t*>(pr + n) = acc[mt];
        if (wp2) *reinterpret_cast<f32x4_t*>(pr + p.swiglu_F + n) = acc2[mt];
      } else {
        f32x4_t v = acc[mt];
        if (wp2) {
#pragma unroll
          for (int i = 0; i < 4; i++) v[i] = swiglu1(v[i], acc2[mt][i]);
        }
        skinny_store(p, m, n, v);
      }
    }
  }
}

// M <= 16: one workgroup = 16 weight rows, its 16 waves split K between them (in-workgroup split-K, LDS reduction in
// fixed order -> deterministic, single launch).  Each wave streams its [16 rows x K/16] slab of W with full 128-B
// lines and reads the matching x fragment straight from L2 (x is M*K*2 bytes = 128 KB for K = 4096: L2 resident).
// N = 4096 alone gives 256 workgroups x 16 waves = 4096 waves of independent loads, enough to cover HBM latency.
constexpr int NW16 = 16;
__global__ __launch_bounds__(NW16 * 64) void gemm_skinny16_kernel(SkinnyParams p) {
  __shared__ float red[NW16][256 + 4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int kw = p.kslice;  // k elements per wave (multiple of 64)
  const int k_begin = wave * kw, k_end = min(p.K, k_begin + kw);
  const bf16_t* wp = skinny_wrow(p, min(n0 + r, p.N - 1)) + g * 16;
  const bf16_t* xp = p.A + (int64_t)min(r, p.M - 1) * p.lda + g * 16;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = k_begin; k0 < k_end; k0 += KC) {
    const int kc = min(KC, k_end - k0);  // multiple of 64
    u16x8_t w[KC / 64][2], x[KC / 64][2];
#pragma unroll
    for (int ks = 0; ks < KC / 64; ks++) {
      if (ks * 64 < kc) {
        w[ks][0] = *reinterpret_cast<const u16x8_t*>(wp + k0 + ks * 64);
        w[ks][1] = *reinterpret_cast<const u16x8_t*>(wp + k0 + ks * 64 + 8);
        x[ks][0] = *reinterpret_cast<const u16x8_t*>(xp + k0 + ks * 64);
        x[ks][1] = *reinterpret_cast<const u16x8_t*>(xp + k0 + ks * 64 + 8);
      }
    }
#pragma unroll
    for (int ks = 0; ks < KC / 64; ks++) {
      if (ks * 64 < kc) {
        acc = mfma16(w[ks][0], x[ks][0], acc);
        acc = mfma16(w[ks][1], x[ks][1], acc);
      }
    }
  }
  // acc[i] = y[m = r][n = n0 + 4g + i]; rows r >= M carry x row M-1 again and are dropped here
  *reinterpret_cast<f32x4_t*>(&red[wave][lane * 4]) = acc;
  __syncthreads();
  if (tid < 64) {  // thread -> (m = tid >> 2, 4 consecutive n)
    const int m = tid >> 2, q = tid & 3;
    const int n = n0 + q * 4;
    if (m < p.M && n < p.N) {
      const int src = (q * 16 + m) * 4;
      f32x4_t v = *reinterpret_cast<const f32x4_t*>(&red[0][src]);
#pragma unroll
      for (int w = 1; w < NW16; w++) {
        const f32x4_t t = *reinterpret_cast<const f32x4_t*>(&red[w][src]);
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] += t[i];
      }
      skinny_store(p, m, n, v);
    }
  }
}

__global__ __launch_bounds__(256) void skinny_reduce_kernel(SkinnyParams p, int S) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;  // float4 index over [M, NO/4]
  const int NO = p.swiglu_F ? p.swiglu_F : p.N;
  const int n4 = NO / 4;
  if (q >= (int64_t)p.M * n4) return;
  const int m = (int)(q / n4), n = (int)(q % n4) * 4;
  f32x4_t v = *reinterpret_cast<const f32x4_t*>(p.part + (int64_t)m * p.N + n);
  for (int s = 1; s < S; s++) {  // fixed summation order -> bit-reproducible
    const f32x4_t t = *reinterpret_cast<const f32x4_t*>(p.part + ((int64_t)s * p.M + m) * p.N + n);
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] += t[i];
  }
  if (p.swiglu_F) {
    f32x4_t u = *reinterpret_cast<const f32x4_t*>(p.part + (int64_t)m * p.N + p.swiglu_F + n);
    for (int s = 1; s < S; s++) {
      const f32x4_t t = *reinterpret_cast<const f32x4_t*>(p.part + ((int64_t)s * p.M + m) * p.N + p.swiglu_F + n);
#pragma unroll
      for (int i = 0; i < 4; i++) u[i] += t[i];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = swiglu1(v[i], u[i]);
  }
  skinny_store(p, m, n, v);
}

void skinny_plan(int64_t N, int64_t K, int64_t splits, int* S, int* kslice) {
  const int64_t blocks_n = cdiv64(N, 64), kchunks = cdiv64(K, KC);
  int64_t s = splits > 0 ? splits : (blocks_n >= 256 ? 1 : cdiv64(1024, blocks_n));
  if (s > kchunks) s = kchunks;
  if (s > 64) s = 64;
  const int64_t ks = cdiv64(kchunks, s) * KC;
  *kslice = (int)ks;
  *S = (int)cdiv64(K, ks);
}

// ------------------------------------------------------------------------------------------------------------
// decode attention, fused with the LoRA delta of the q/k/v projections, RoPE and the KV append of the new token.
// grid = (Hkv, rows); the REP q heads of a kv head share its K/V stream.
// ------------------------------------------------------------------------------------------------------------
struct DecodeAttnParams {
  const bf16_t* qkv; int64_t ld;   // [R, (Hq+2Hkv)*D (+ lora_r)]: new token's q|k|v (pre-RoPE) (| u = x A^T)
  const bf16_t* lora_b; int64_t ldlb; int lora_r;  // packed (alpha/r) B: [(Hq+2Hkv)*D, >= lora_r] or null
  const float* cosT; const float* sinT; const int* positions;  // RoPE tables [P, D/2], position per row
  const bf16_t* Kp; const bf16_t* Vp;  // prompt KV [B, Tp, Hkv*D]
  const int* start;                // [B] first valid prompt slot (left padding)
  bf16_t* Kg; bf16_t* Vg;          // generated KV [R, G, Hkv*D]
  int* anc;                        // [R, G] physical row that wrote slot j of hypothesis r
  const int* gen_count_dev; int gen_count_host;  // generated tokens cached BEFORE this step's append
  bf16_t* O; int64_t ldo;
  int beams, Tp, G, Hq, Hkv;
  float scale_log2;
};

constexpr int MAXG = 1024;

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the LPK (8 or 16) consecutive lanes of a key group, result in every lane of the group (DPP only)
template <int LPK>
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);  // row_half_mirror
  if constexpr (LPK == 16) v += dpp_f<0x140>(v);  // row_mirror
  return v;
}
// value of lane ^ OFF for OFF = 8 (DPP row rotate), 16 / 32 (gfx950 v_permlane16_swap / v_permlane32_swap)
template <int OFF>
__device__ __forceinline__ float lane_xor(float v, int lane) {
  if constexpr (OFF == 8) {
    return dpp_f<0x128>(v);  // row_ror:8
  } else {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    if constexpr (OFF == 16) {
      const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
      return __builtin_bit_cast(float, (lane & 16) ? r[0] : r[1]);
    } else {
      const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
      return __builtin_bit_cast(float, (lane & 32) ? r[0] : r[1]);
    }
  }
}

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

template <int REP>
struct AttnState {
  float m[REP], l[REP];
  f32x2_t acc[REP][4];   // 8 head-dim elements per lane as 4 packed pairs (v_pk_fma_f32 / v_pk_mul_f32)
};

// NU keys at once for the REP query heads sharing this K/V stream.  QK^T on packed bf16 pairs (v_dot2c_f32_bf16, no
// bf16->f32 conversion of q or k), one running-max update and one accumulator rescale per NU keys instead of per key,
// PV on packed fp32 pairs.  valid[u] = false -> the key contributes nothing (at least one key must be valid).
template <int REP, int LPK, int NU>
__device__ __forceinline__ void attn_keys(AttnState<REP>& st, const u16x8_t (&qraw)[REP], const u16x8_t* kk,
                                          const u16x8_t* vv, const bool* valid, float scale_log2) {
  float sc[NU][REP];
#pragma unroll
  for (int u = 0; u < NU; u++) {
    const u32x4_t kp = __builtin_bit_cast(u32x4_t, kk[u]);
#pragma unroll
    for (int h = 0; h < REP; h++) {
      const u32x4_t qp = __builtin_bit_cast(u32x4_t, qraw[h]);
      // s = sum over the lane's 4 packed bf16 pairs of q.lo*k.lo + q.hi*k.hi.  Inline asm on purpose: with hipcc 7.2 the
      // builtin (__builtin_amdgcn_fdot2_f32_bf16 on bit-cast vector lanes) folds all four pairs onto the first pair's
      // registers (4 identical v_dot2c in the ISA, wrong scores; tests/test_ops_gpu.py::test_decode_attention_fused
      // guards it).  The trailing s_nop covers the VALU-write -> DPP-read hazard of group_sum, which the hazard
      // recogniser cannot see through an asm block.
      float s = 0.f;
      asm("v_dot2c_f32_bf16 %0, %1, %5\n\tv_dot2c_f32_bf16 %0, %2, %6\n\tv_dot2c_f32_bf16 %0, %3, %7\n\t"
          "v_dot2c_f32_bf16 %0, %4, %8\n\ts_nop 1"
          : "+v"(s)
          : "v"(qp[0]), "v"(qp[1]), "v"(qp[2]), "v"(qp[3]), "v"(kp[0]), "v"(kp[1]), "v"(kp[2]), "v"(kp[3]));
      s = group_sum<LPK>(s) * scale_log2;
      sc[u][h] = valid[u] ? s : -INFINITY;
    }
  }
#pragma unroll
  for (int h = 0; h < REP; h++) {
    float mx = st.m[h];
#pragma unroll
    for (int u = 0; u < NU; u++) mx = fmaxf(mx, sc[u][h]);
    const float corr = __builtin_amdgcn_exp2f(st.m[h] - mx);  // m = -inf before the first key -> 0
    float pr[NU], psum = 0.f;
#pragma unroll
    for (int u = 0; u < NU; u++) {
      pr[u] = __builtin_amdgcn_exp2f(sc[u][h] - mx);
      psum += pr[u];
    }
    st.l[h] = st.l[h] * corr + psum;
    const f32x2_t c2 = {corr, corr};
#pragma unroll
    for (int j = 0; j < 4; j++) st.acc[h][j] = st.acc[h][j] * c2;
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const u32x4_t vp = __builtin_bit_cast(u32x4_t, vv[u]);
      const f32x2_t p2 = {pr[u], pr[u]};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const f32x2_t v2 = {__uint_as_float(vp[j] << 16), __uint_as_float(vp[j] & 0xFFFF0000u)};
        st.acc[h][j] = __builtin_elementwise_fma(p2, v2, st.acc[h][j]);
      }
    }
    st.m[h] = mx;
  }
}

template <int REP, int OFF>
__device__ __forceinline__ void attn_merge(AttnState<REP>& st, int lane) {
#pragma unroll
  for (int h = 0; h < REP; h++) {
    const float mo = lane_xor<OFF>(st.m[h], lane), lo = lane_xor<OFF>(st.l[h], lane);
    const float mn = fmaxf(st.m[h], mo);
    const float a = (st.m[h] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(st.m[h] - mn);
    const float b = (mo == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mo - mn);
    st.l[h] = st.l[h] * a + lo * b;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const f32x2_t o = {lane_xor<OFF>(st.acc[h][j][0], lane), lane_xor<OFF>(st.acc[h][j][1], lane)};
      st.acc[h][j] = st.acc[h][j] * f32x2_t{a, a} + o * f32x2_t{b, b};
    }
    st.m[h] = mn;
  }
}

template <int D, int REP>
__global__ __launch_bounds__(1024) void attn_decode_kernel(DecodeAttnParams p) {
  constexpr int NW = 16;         // waves per workgroup: the key stream of one (row, kv head) is short, spread it wide
  constexpr int LPK = D / 8;     // lanes per key row (16 B each)
  constexpr int GPW = 64 / LPK;  // key groups per wave
  constexpr int NG = NW * GPW;   // key groups per workgroup
  constexpr int U = 4;           // keys in flight per group
  constexpr int NH = REP + 2;    // this workgroup's heads of the new token: REP q heads, k, v
  __shared__ int anc_s[MAXG];
  __shared__ __attribute__((aligned(16))) bf16_t tok[NH][D];   // new token: q heads and k (RoPE'd), v -- bf16 as stored
  __shared__ float red_m[NW][REP], red_l[NW][REP];
  __shared__ __attribute__((aligned(16))) float red_acc[NW][REP][D];
  const int hk = blockIdx.x, r = blockIdx.y, item = r / p.beams;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane % LPK, gid = wave * GPW + lane / LPK;
  const int n_prev = p.gen_count_dev ? *p.gen_count_dev : p.gen_count_host;  // slot of the new token
  const int start = p.start ? p.start[item] : 0;
  const int np = p.Tp - start;
  const int n_keys = np + n_prev;   // cached keys; the new token is handled from LDS by group 0
  const int64_t HD = (int64_t)p.Hkv * D;
  for (int j = tid; j < n_prev; j += NW * 64) anc_s[j] = p.anc[(int64_t)r * p.G + j];
  if (tid == 0 && hk == 0) p.anc[(int64_t)r * p.G + n_prev] = r;

  // ---- new token: y = qkv (+ u . (sB)^T), RoPE on q and k, append k/v ------------------------------------
  {
    const bf16_t* row = p.qkv + (int64_t)r * p.ld;
    const int QD = p.Hq * D, KD = p.Hkv * D;
    const int pos = p.positions[r];
    for (int idx = tid; idx < NH * (D / 2); idx += NW * 64) {
      const int hh = idx / (D / 2), d = idx % (D / 2);
      // column of head hh in the fused q|k|v output
      const int col = hh < REP ? (hk * REP + hh) * D : (hh == REP ? QD + hk * D : QD + KD + hk * D);
      float x1 = bf2f(row[col + d]), x2 = bf2f(row[col + D / 2 + d]);
      if (p.lora_b) {   // peft: result = base(x) + scale * B(A(x)), both terms rounded to bf16 before the add
        const bf16_t* u = row + QD + 2 * KD;
        const bf16_t* b1 = p.lora_b + (int64_t)(col + d) * p.ldlb;
        const bf16_t* b2 = p.lora_b + (int64_t)(col + D / 2 + d) * p.ldlb;
        float d1 = 0.f, d2 = 0.f;
        int j = 0;
        if ((p.ldlb & 7) == 0 && (p.ld & 7) == 0) {  // 16-byte aligned rows: 8 ranks per load
          for (; j + 8 <= p.lora_r; j += 8) {
            const u16x8_t uv = *reinterpret_cast<const u16x8_t*>(u + j);
            const u16x8_t v1 = *reinterpret_cast<const u16x8_t*>(b1 + j);
            const u16x8_t v2 = *reinterpret_cast<const u16x8_t*>(b2 + j);
#pragma unroll
            for (int e = 0; e < 8; e++) {
              d1 = fmaf(bf2f(uv[e]), bf2f(v1[e]), d1);
              d2 = fmaf(bf2f(uv[e]), bf2f(v2[e]), d2);
            }
          }
        }
        for (; j < p.lora_r; j++) {
          const float uj = bf2f(u[j]);
          d1 = fmaf(uj, bf2f(b1[j]), d1);
          d2 = fmaf(uj, bf2f(b2[j]), d2);
        }
        x1 = bf2f(f2bf(x1 + bf2f(f2bf(d1))));
        x2 = bf2f(f2bf(x2 + bf2f(f2bf(d2))));
      }
      if (hh <= REP) {  // q heads and k: rotate (HF rotate_half convention), round to bf16 like the prefill path
        const float cs = p.cosT[(int64_t)pos * (D / 2) + d], sn = p.sinT[(int64_t)pos * (D / 2) + d];
        const float y1 = bf2f(f2bf(x1 * cs - x2 * sn)), y2 = bf2f(f2bf(x2 * cs + x1 * sn));
        x1 = y1;
        x2 = y2;
      }
      if (hh >= REP) {
        bf16_t* dst = (hh == REP ? p.Kg : p.Vg) + ((int64_t)r * p.G + n_prev) * HD + hk * D;
        dst[d] = f2bf(x1);
        dst[D / 2 + d] = f2bf(x2);
      }
      tok[hh][d] = f2bf(x1);
      tok[hh][D / 2 + d] = f2bf(x2);
    }
  }
  __syncthreads();

  u16x8_t q[REP];
#pragma unroll
  for (int h = 0; h < REP; h++) q[h] = *reinterpret_cast<const u16x8_t*>(&tok[h][c * 8]);
  AttnState<REP> st;
#pragma unroll
  for (int h = 0; h < REP; h++) {
    st.m[h] = -INFINITY;
    st.l[h] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) st.acc[h][j] = f32x2_t{0.f, 0.f};
  }
  if (gid == 0) {  // the new token itself
    const u16x8_t k1 = *reinterpret_cast<const u16x8_t*>(&tok[REP][c * 8]);
    const u16x8_t v1 = *reinterpret_cast<const u16x8_t*>(&tok[REP + 1][c * 8]);
    const bool ok = true;
    attn_keys<REP, LPK, 1>(st, q, &k1, &v1, &ok, p.scale_log2);
  }
  const bf16_t* kp_base = p.Kp + ((int64_t)item * p.Tp + start) * HD + hk * D + c * 8;
  const bf16_t* vp_base = p.Vp + ((int64_t)item * p.Tp + start) * HD + hk * D + c * 8;
  for (int i0 = gid; i0 < n_keys; i0 += NG * U) {
    u16x8_t kk[U], vv[U];
    bool valid[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = i0 + u * NG;
      valid[u] = i < n_keys;
      kk[u] = vv[u] = u16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
      if (valid[u]) {
        int64_t off;
        const bf16_t *kb, *vb;
        if (i < np) {
          off = (int64_t)i * HD;
          kb = kp_base;
          vb = vp_base;
        } else {
          const int j = i - np;
          off = ((int64_t)anc_s[j] * p.G + j) * HD + hk * D + c * 8;
          kb = p.Kg;
          vb = p.Vg;
        }
        kk[u] = *reinterpret_cast<const u16x8_t*>(kb + off);
        vv[u] = *reinterpret_cast<const u16x8_t*>(vb + off);
      }
    }
    attn_keys<REP, LPK, U>(st, q, kk, vv, valid, p.scale_log2);   // valid[0] is always true here
  }
  // merge the key groups of a wave (lanes with equal c) in registers, then the waves through LDS
  if constexpr (LPK == 8) attn_merge<REP, 8>(st, lane);
  attn_merge<REP, 16>(st, lane);
  attn_merge<REP, 32>(st, lane);
  if (lane < LPK) {
#pragma unroll
    for (int h = 0; h < REP; h++) {
      if (c == 0) {
        red_m[wave][h] = st.m[h];
        red_l[wave][h] = st.l[h];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        red_acc[wave][h][c * 8 + 2 * j] = st.acc[h][j][0];
        red_acc[wave][h][c * 8 + 2 * j + 1] = st.acc[h][j][1];
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < REP * D; idx += NW * 64) {
    const int h = idx / D, d = idx % D;
    float mx = red_m[0][h];
#pragma unroll
    for (int w = 1; w < NW; w++) mx = fmaxf(mx, red_m[w][h]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const float a = (red_m[w][h] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(red_m[w][h] - mx);
      num += a * red_acc[w][h][d];
      den += a * red_l[w][h];
    }
    p.O[(int64_t)r * p.ldo + (hk * REP + h) * D + d] = f2bf(num / den);
  }
}

template <int D>
int launch_attn_decode(const DecodeAttnParams& p, int rep, int R, hipStream_t st) {
  dim3 grid((unsigned)p.Hkv, (unsigned)R);
  switch (rep) {
    case 1: hipLaunchKernelGGL((attn_decode_kernel<D, 1>), grid, dim3(1024), 0, st, p); break;
    case 2: hipLaunchKernelGGL((attn_decode_kernel<D, 2>), grid, dim3(1024), 0, st, p); break;
    case 4: hipLaunchKernelGGL((attn_decode_kernel<D, 4>), grid, dim3(1024), 0, st, p); break;
    case 8: hipLaunchKernelGGL((attn_decode_kernel<D, 8>), grid, dim3(1024), 0, st, p); break;
    default: return -1;
  }
  return 0;
}

}  // namespace

extern "C" int64_t slam_gemm_skinny_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t splits, int swiglu) {
  if (M <= 16 && !swiglu && (N < 16384 || splits < 0)) return 0;
  int S, ks;
  skinny_plan(swiglu ? N / 2 : N, K, splits, &S, &ks);
  return S > 1 ? (int64_t)S * M * N * 4 : 0;
}

extern "C" int slam_gemm_skinny_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, const void* B2,
                                        int64_t ldb2, int64_t N2, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                        const void* residual, int64_t ldr, int out_dtype, int swiglu, void* workspace,
                                        int64_t workspace_bytes, int64_t splits, void* stream) {
  SLAM_CHECK_ARG(A && B && C, "slam_gemm_skinny_bf16_nt: null pointer");
  SLAM_CHECK_ARG(M > 0 && M <= 64, "slam_gemm_skinny_bf16_nt: M must be in [1, 64] (got %lld)", (long long)M);
  SLAM_CHECK_ARG(N > 0 && N % 4 == 0 && K > 0 && K % 64 == 0, "slam_gemm_skinny_bf16_nt: need N %% 4 == 0 and K %% 64 == 0");
  SLAM_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && (!residual || ldr % 4 == 0),
                 "slam_gemm_skinny_bf16_nt: leading dimensions must keep 16-byte (A, B) / 8-byte (C, residual) alignment");
  SLAM_CHECK_ARG(out_dtype == SLAM_BF16 || out_dtype == SLAM_F32, "slam_gemm_skinny_bf16_nt: bad out_dtype");
  SLAM_CHECK_ARG((!B2 && N2 == 0) || (B2 && N2 > 0 && N2 % 4 == 0 && ldb2 % 8 == 0), "slam_gemm_skinny_bf16_nt: bad second weight block");
  SLAM_CHECK_ARG(!swiglu || (N % 8 == 0 && !B2 && !residual), "slam_gemm_skinny_bf16_nt: swiglu needs N = 2F, no B2, no residual");
  hipStream_t st = (hipStream_t)stream;
  const int64_t Nt = N + N2;  // total weight rows
  SkinnyParams p;
  p.A = (const bf16_t*)A; p.lda = lda; p.B = (const bf16_t*)B; p.ldb = ldb;
  p.B2 = (const bf16_t*)B2; p.ldb2 = ldb2; p.N1 = (int)N; p.swiglu_F = swiglu ? (int)(N / 2) : 0;
  p.part = nullptr;
  p.C = C; p.ldc = ldc; p.res = (const bf16_t*)residual; p.ldr = ldr;
  p.M = (int)M; p.N = (int)Nt; p.K = (int)K; p.out_f32 = out_dtype == SLAM_F32;
  // M <= 16: in-workgroup split-K unless N alone already fills the chip (then x is better shared through LDS by the
  // 4 waves of the 64-row kernel: measured 4.5-5.3 TB/s vs 3.7-4.1 TB/s at N = 28672 / 128256)
  if (M <= 16 && !swiglu && (Nt < 16384 || splits < 0)) {
    p.kslice = (int)(cdiv64(cdiv64(K, 64), NW16) * 64);
    hipLaunchKernelGGL(gemm_skinny16_kernel, dim3((unsigned)cdiv64(Nt, 16)), dim3(NW16 * 64), 0, st, p);
    SLAM_CHECK_LAUNCH("slam_gemm_skinny_bf16_nt(M<=16)");
    return 0;
  }
  int S, ks;
  skinny_plan(swiglu ? N / 2 : Nt, K, splits, &S, &ks);
  SLAM_CHECK_ARG(S == 1 || (workspace && workspace_bytes >= (int64_t)S * M * Nt * 4),
                 "slam_gemm_skinny_bf16_nt: workspace too small (need %lld bytes)", (long long)((int64_t)S * M * Nt * 4));
  p.part = S > 1 ? (float*)workspace : nullptr;
  p.kslice = ks;
  const int64_t NO = swiglu ? N / 2 : Nt;
  dim3 grid((unsigned)cdiv64(NO, 64), (unsigned)S);
  const int mt = (int)cdiv64(M, 16);
  if (mt == 1) hipLaunchKernelGGL(gemm_skinny_kernel<1>, grid, dim3(256), 0, st, p);
  else if (mt == 2) hipLaunchKernelGGL(gemm_skinny_kernel<2>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(gemm_skinny_kernel<4>, grid, dim3(256), 0, st, p);
  SLAM_CHECK_LAUNCH("slam_gemm_skinny_bf16_nt");
  if (S > 1) {
    hipLaunchKernelGGL(skinny_reduce_kernel, dim3((unsigned)cdiv64(M * (NO / 4), 256)), dim3(256), 0, st, p, S);
    SLAM_CHECK_LAUNCH("slam_gemm_skinny_bf16_nt(reduce)");
  }
  return 0;
}

extern "C" int slam_attn_decode(const void* qkv, int64_t ld, const void* lora_b, int64_t ldlb, int64_t lora_r,
                                const float* cos_table, const float* sin_table, const int32_t* positions,
                                const void* k_prompt, const void* v_prompt, const int32_t* prompt_start, void* k_gen,
                                void* v_gen, int32_t* ancestors, const int32_t* gen_count_dev, int64_t gen_count, void* O,
                                int64_t ldo, int64_t R, int64_t beams, int64_t Tp, int64_t G, int64_t Hq, int64_t Hkv,
                                int64_t D, float scale, void* stream) {
  SLAM_CHECK_ARG(qkv && cos_table && sin_table && positions && k_prompt && v_prompt && k_gen && v_gen && ancestors && O,
                 "slam_attn_decode: null pointer");
  SLAM_CHECK_ARG(D == 64 || D == 128, "slam_attn_decode: head_dim must be 64 or 128 (got %lld)", (long long)D);
  SLAM_CHECK_ARG(R > 0 && beams > 0 && R % beams == 0 && Tp >= 0 && G > 0 && G <= MAXG,
                 "slam_attn_decode: bad sizes (generated-token capacity must be <= %d)", MAXG);
  SLAM_CHECK_ARG(gen_count_dev || (gen_count >= 0 && gen_count < G), "slam_attn_decode: generated-token cache full");
  SLAM_CHECK_ARG(Hkv > 0 && Hq % Hkv == 0, "slam_attn_decode: Hq must be a multiple of Hkv");
  SLAM_CHECK_ARG(!lora_b || (lora_r > 0 && ldlb >= lora_r), "slam_attn_decode: bad LoRA block");
  DecodeAttnParams p;
  p.qkv = (const bf16_t*)qkv; p.ld = ld; p.lora_b = (const bf16_t*)lora_b; p.ldlb = ldlb; p.lora_r = (int)lora_r;
  p.cosT = cos_table; p.sinT = sin_table; p.positions = positions;
  p.Kp = (const bf16_t*)k_prompt; p.Vp = (const bf16_t*)v_prompt; p.start = prompt_start;
  p.Kg = (bf16_t*)k_gen; p.Vg = (bf16_t*)v_gen; p.anc = ancestors;
  p.gen_count_dev = gen_count_dev; p.gen_count_host = (int)gen_count; p.O = (bf16_t*)O; p.ldo = ldo;
  p.beams = (int)beams; p.Tp = (int)Tp; p.G = (int)G; p.Hq = (int)Hq; p.Hkv = (int)Hkv;
  p.scale_log2 = scale * 1.44269504088896340736f;
  const int rc = D == 64 ? launch_attn_decode<64>(p, (int)(Hq / Hkv), (int)R, (hipStream_t)stream)
                         : launch_attn_decode<128>(p, (int)(Hq / Hkv), (int)R, (hipStream_t)stream);
  SLAM_CHECK_ARG(rc == 0, "slam_attn_decode: Hq/Hkv must be 1, 2, 4 or 8 (got %lld)", (long long)(Hq / Hkv));
  SLAM_CHECK_LAUNCH("slam_attn_decode");
  return 0;
}
