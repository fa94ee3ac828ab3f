// Shared by the bf16 GEMM code (gemm_bf16.hip; round 5 had a second translation unit): launch parameters, LDS-DMA helpers and the epilogues.
// Split out of gemm_bf16.hip in round 5 so that a new kernel form compiles in its own (short) translation unit.
#pragma once
#include "common.h"
#include <type_traits>

struct GemmParams {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  int64_t lda, ldb, ldc;
  int M, N, K;
  const float* bias;
  const bf16_t* res;
  int64_t ldr;
  int res_mod;
  int act;  // 0 none, 1 gelu(erf), 2 relu, 3 swiglu backward (res = [gate | up] of the forward, N = F), 4 swiglu forward (C2 = h)
  void* C2;      // act 4: second output h[M, N/2] = silu(gate) * up (bf16)
  int64_t ldc2;
  float alpha;
  int out_f32;
  int accumulate;
  int tiles_m, tiles_n;
  int group_m;   // M-tiles per raster group (L2 reuse of the B panel inside an XCD)
  // split-K tail of the 4-wave kernel (sk_S >= 2; see gemm_nt_w4_kernel): the first sk_main workgroups compute whole tiles, the
  // other sk_R * sk_S compute 1 / sk_S of the K range of one of the last sk_R tiles each
  int sk_main, sk_R, sk_S;
  float* sk_ws;        // [sk_R * sk_S] fp32 slabs of 256 x 256 partial sums, lane-linear
  unsigned* sk_cnt;    // [sk_R] arrival counters, zero between launches
};

// (XCD, entry of that XCD) -> position in the tile order.  The hardware deals workgroups to the 8 XCDs round-robin, so XCD x receives entries
// 0 .. q (- 1) with q = nwg / 8 (the first nwg % 8 XCDs one more); any bijection of those pairs onto [0, nwg) is a valid order.  Shipped: every XCD
// owns ONE contiguous run.  (Round 6 measured the block-cyclic alternatives in the C3 step -- XCD x owning blocks x, x + 8, ... of one raster group /
// 256 / 32 / 1 positions: +0.6 % / +0.25 % / +1.2 % / +5.1 % step time; the per-XCD L2 locality of a long contiguous run is what pays.  profiles/r06_gemm_raster.md)
__device__ __forceinline__ int gemm_order_pos(const GemmParams& p, int nwg, int xcd, int idx) {
  const int q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// position in the tile order -> (tile row, tile column): groups of group_m M-tiles x all N-tiles, M fastest; scalar arithmetic (the position is
// workgroup-uniform).  (Round 6 also tried column BANDS in this order -- every XCD owning ~1 / xc of the columns and ~xc / 8 of the rows, which
// halves the launch's COMPULSORY fabric traffic xc |A| + (8 / xc) |B| for the wide products -- and measured it 0.65 % SLOWER in the C3 step:
// with one band all XCDs sweep the same B columns at the same time, and that alignment is worth more than the byte count; profiles/r06_gemm_raster.md.)
__device__ __forceinline__ void gemm_tile_of(const GemmParams& p, int bid, int& tm, int& tn) {
  const int GM = p.group_m;
  const int per_group = GM * p.tiles_n;
  const int group = bid / per_group;
  const int first_m = group * GM;
  const int gsz = min(GM, p.tiles_m - first_m);
  const int within = bid - group * per_group;
  tm = first_m + within % gsz;
  tn = within / gsz;
}

constexpr int BK = 64;           // bf16 elements per K-tile
constexpr int ROWB = BK * 2;     // 128 bytes per LDS row

__device__ __forceinline__ void glds16(const bf16_t* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// LDS-DMA through a buffer descriptor (buffer_load_dwordx4 ... offen lds): 32-bit per-lane byte offset, range-checked
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t srd, unsigned voff, char* l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)l, 16, (int)voff, 0, 0, 0);
}

// swap the odd 16-lane rows of `a` with the even rows of `b` (gfx950).  Inline asm: this hipcc folds the builtin's second result
// into its first, and the two operands must be different registers.
__device__ __forceinline__ void swap_rows16(unsigned& a, unsigned& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

template <int FM, int FN, int WTM, int WTN>
__device__ __forceinline__ void gemm_epilogue_generic(const GemmParams& p, f32x4_t (&acc)[FM][FN], int m0, int n0, int wm,
                                              int wn, int frow, int fg) {
  // ---- epilogue: lane holds C[m = ..+lane&15][n = ..+(lane>>4)*4 + 0..3] of every 16x16 fragment ----
  // The store tail of a tile is ISSUE bound (one 256x256 bf16 tile = 128 KiB: ~19 k cycles with 8-byte stores, the time of ten
  // k-tiles -- 14 % of a K = 4096 tile, a third of a K = 1280 one).  bf16 outputs therefore leave in 16-byte stores: the packed
  // halves of two neighbouring fragments are exchanged between lane rows (v_permlane16_swap: odd rows of the first with even rows
  // of the second), after which a lane owns EIGHT consecutive columns of fragment 2*jp + (fg & 1), starting at column (fg >> 1) * 8.
  static_assert(FN % 2 == 0, "fragments are stored in pairs");
  const bool wide = !p.out_f32 && !p.accumulate;
#pragma unroll
  for (int i = 0; i < FM; i++) {
    const int m = m0 + wm * WTM + i * 16 + frow;
    if (m >= p.M) continue;   // (both lanes of an exchanging pair share frow, hence m)
#pragma unroll
    for (int jp = 0; jp < FN / 2; jp++) {
      unsigned pk[2][2] = {{0u, 0u}, {0u, 0u}};
#pragma unroll
      for (int hh = 0; hh < 2; hh++) {
        const int j = 2 * jp + hh;
        const int n = n0 + wn * WTN + j * 16 + fg * 4;
        if (n >= p.N) continue;  // N % 4 == 0 is enforced by the launcher
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = acc[i][j][e] * p.alpha;
        if (p.bias) {
          const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (p.act == 1) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = gelu_erf(v[e]);
        } else if (p.act == 2) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.act == 3) {
          // SwiGLU backward fused into the down_proj dX GEMM: acc = dL/dh for h = silu(gate) * up.  Reads the forward's
          // gate/up (p.res, columns n and N + n), writes dL/dgate to C[m, n] (below) and dL/dup to C[m, N + n] -- the
          // [M, F] intermediate dL/dh and its elementwise pass never touch HBM.  Same arithmetic as swiglu_bwd_kernel
          // (elementwise.hip) applied to the bf16-rounded product.
          const u16x4_t g4 = *reinterpret_cast<const u16x4_t*>(p.res + (int64_t)m * p.ldr + n);
          const u16x4_t u4 = *reinterpret_cast<const u16x4_t*>(p.res + (int64_t)m * p.ldr + p.N + n);
          float du[4];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float gf = bf2f(g4[e]), uf = bf2f(u4[e]), df = bf2f(f2bf(v[e]));
            const float sg = sigmoid_fast(gf);
            v[e] = df * uf * sg * (1.0f + gf * (1.0f - sg));
            du[e] = df * (gf * sg);
          }
          uint2 o2;
          o2.x = pack2bf(du[0], du[1]);
          o2.y = pack2bf(du[2], du[3]);
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + p.N + n) = o2;
        } else if (p.res) {
          const int rr = p.res_mod > 0 ? (m % p.res_mod) : m;
          const u16x4_t r4 = *reinterpret_cast<const u16x4_t*>(p.res + (int64_t)rr * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] += bf2f(r4[e]);
        }
        if (p.out_f32) {
          float* c = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n;
          if (p.accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(c);
            v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
          }
          *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + n;
          if (p.accumulate) {
            const u16x4_t o = *reinterpret_cast<const u16x4_t*>(c);
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] += bf2f(o[e]);
          }
          pk[hh][0] = pack2bf(v[0], v[1]);
          pk[hh][1] = pack2bf(v[2], v[3]);
          if (!wide) {
            uint2 o2;
            o2.x = pk[hh][0];
            o2.y = pk[hh][1];
            *reinterpret_cast<uint2*>(c) = o2;
          }
        }
      }
      if (wide) {
        swap_rows16(pk[0][0], pk[1][0]);
        swap_rows16(pk[0][1], pk[1][1]);
        const int nn = n0 + wn * WTN + (2 * jp + (fg & 1)) * 16 + (fg >> 1) * 8;
        bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + nn;
        if (nn + 8 <= p.N) {
          *reinterpret_cast<uint4*>(c) = make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
        } else if (nn < p.N) {
          uint2 o2;
          o2.x = pk[0][0];
          o2.y = pk[0][1];
          *reinterpret_cast<uint2*>(c) = o2;
        }
      }
    }
  }
}

// ---- epilogue: lane holds C[m = ..+lane&15][n = ..+(lane>>4)*4 + 0..3] of every 16x16 fragment ----
// The epilogue must be SHORT CODE.  The first form handled every option (bias, GELU / ReLU / SwiGLU-backward, residual, fp32 /
// accumulate) with wave-uniform branches inside the unrolled fragment loop: 12 000 lines of ISA and 650 branches per kernel,
// through which the one live path hopped from cache line to cache line -- 24 000 cycles per 256x256 tile measured with a single
// workgroup alone on the chip (tools/gemm_epi_probe.py), i.e. the time of ten k-tiles: 13 % of a K = 4096 tile, 27 % of a
// K = 1280 one.  Now one wave-uniform switch picks a specialised straight-line loop (bias x activation x residual known at
// compile time; bf16 output in 16-byte stores); the everything-else form is kept out of line for fp32 / accumulating /
// SwiGLU-backward products.
template <int FM, int FN, int WTM, int WTN, bool BIAS, int ACT, bool RES, bool INNER>
__device__ __forceinline__ void gemm_epilogue_bf16(const GemmParams& p, f32x4_t (&acc)[FM][FN], int m0, int n0, int wm, int wn,
                                                   int frow, int fg) {
  static_assert(FN % 2 == 0, "fragments are stored in pairs");
  // bf16 outputs leave in 16-byte stores: the packed halves of two neighbouring fragments are exchanged between lane rows
  // (v_permlane16_swap: odd rows of the first with even rows of the second), after which a lane owns EIGHT consecutive columns of
  // fragment 2*jp + (fg & 1), starting at column (fg >> 1) * 8.
  // INNER: the workgroup's whole tile lies inside [0, M) x [0, N) -- no per-lane guards (all but the last tile row / column).
  // A row's residual values are requested together, ahead of its arithmetic (one wait per row, not one per fragment).
  // (alpha == 1 here: gemm_epilogue sends scaled products through the generic form -- as a wave-uniform `if` inside this loop the
  // compiler turned the scaling into 4 v_pk_mul + 8 v_cndmask per 16-byte store, half the VALU of the plain path; and with ONE wave
  // per SIMD the store tail of the 4-wave kernel is bound by exactly that dependent VALU chain, ~350 cycles per store)
  const int nbase = n0 + wn * WTN;
  float4 bias4[FN];
  if constexpr (BIAS) {
#pragma unroll
    for (int j = 0; j < FN; j++) {
      const int n = nbase + j * 16 + fg * 4;
      bias4[j] = (INNER || n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  bf16_t* const crow0 = reinterpret_cast<bf16_t*>(p.C) + (int64_t)(m0 + wm * WTM + frow) * p.ldc + nbase + (fg & 1) * 16 + (fg >> 1) * 8;
#pragma unroll
  for (int i = 0; i < FM; i++) {
    const int m = m0 + wm * WTM + i * 16 + frow;
    if (!INNER && m >= p.M) continue;   // (both lanes of an exchanging pair share frow, hence m)
    u16x4_t res4[FN];
    if constexpr (RES) {
      const bf16_t* rrow = p.res + (int64_t)(p.res_mod > 0 ? (m % p.res_mod) : m) * p.ldr + nbase + fg * 4;
#pragma unroll
      for (int j = 0; j < FN; j++) {
        if (INNER || nbase + j * 16 + fg * 4 < p.N) res4[j] = *reinterpret_cast<const u16x4_t*>(rrow + j * 16);
        else res4[j] = u16x4_t{0, 0, 0, 0};
      }
    }
    bf16_t* crow = crow0 + (int64_t)i * 16 * p.ldc;   // (one 64-bit add per row, not a 64-bit multiply)
    // all packed values of the row first, then its stores back to back: with the 16 bytes of every store in the SAME four registers
    // each conversion had to wait until the previous store had read its data out of the register file
    uint4 outv[FN / 2];
#pragma unroll
    for (int jp = 0; jp < FN / 2; jp++) {
      unsigned pk[2][2];
#pragma unroll
      for (int hh = 0; hh < 2; hh++) {
        const int j = 2 * jp + hh;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = acc[i][j][e];
        if constexpr (BIAS) {
          v[0] += bias4[j].x; v[1] += bias4[j].y; v[2] += bias4[j].z; v[3] += bias4[j].w;
        }
        if constexpr (ACT == 1) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = gelu_erf(v[e]);
        } else if constexpr (ACT == 2) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (RES) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] += bf2f(res4[j][e]);
        }
        pk[hh][0] = pack2bf(v[0], v[1]);
        pk[hh][1] = pack2bf(v[2], v[3]);
      }
      swap_rows16(pk[0][0], pk[1][0]);
      swap_rows16(pk[0][1], pk[1][1]);
      outv[jp] = make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
    }
#pragma unroll
    for (int jp = 0; jp < FN / 2; jp++) {
      bf16_t* c = crow + jp * 32;
      if constexpr (INNER) {
        *reinterpret_cast<uint4*>(c) = outv[jp];
      } else {
        const int nn = nbase + (2 * jp + (fg & 1)) * 16 + (fg >> 1) * 8;
        if (nn + 8 <= p.N) {
          *reinterpret_cast<uint4*>(c) = outv[jp];
        } else if (nn < p.N) {   // N % 4 == 0 is enforced by the launcher
          uint2 o2;
          o2.x = outv[jp].x;
          o2.y = outv[jp].y;
          *reinterpret_cast<uint2*>(c) = o2;
        }
      }
    }
  }
}

// SwiGLU backward fused into the down_proj dX product (act 3): acc = dL/dh for h = silu(gate) * up; p.res = the forward's
// [gate | up] ([M, 2N]); writes dL/dgate to C[m, n] and dL/dup to C[m, N + n] -- the [M, F] intermediate and its elementwise pass
// (1.7 GB of traffic per layer at the Llama shape) never touch HBM.  Same arithmetic as swiglu_bwd_kernel (elementwise.hip) on the
// bf16-rounded product; same 16-byte store scheme as gemm_epilogue_bf16, for both outputs.
template <int FM, int FN, int WTM, int WTN, bool INNER>
__device__ __forceinline__ void gemm_epilogue_swiglu_bwd(const GemmParams& p, f32x4_t (&acc)[FM][FN], int m0, int n0, int wm, int wn,
                                                         int frow, int fg) {
  static_assert(FN % 2 == 0, "fragments are stored in pairs");
  const int nbase = n0 + wn * WTN;
#pragma unroll
  for (int i = 0; i < FM; i++) {
    const int m = m0 + wm * WTM + i * 16 + frow;
    if (!INNER && m >= p.M) continue;
    const bf16_t* grow = p.res + (int64_t)m * p.ldr + nbase + fg * 4;
    u16x4_t g4[FN], u4[FN];
#pragma unroll
    for (int j = 0; j < FN; j++) {
      if (INNER || nbase + j * 16 + fg * 4 < p.N) {
        g4[j] = *reinterpret_cast<const u16x4_t*>(grow + j * 16);
        u4[j] = *reinterpret_cast<const u16x4_t*>(grow + p.N + j * 16);
      } else {
        g4[j] = u4[j] = u16x4_t{0, 0, 0, 0};
      }
    }
    bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + nbase + (fg & 1) * 16 + (fg >> 1) * 8;
#pragma unroll
    for (int jp = 0; jp < FN / 2; jp++) {
      unsigned pg[2][2], pu[2][2];
#pragma unroll
      for (int hh = 0; hh < 2; hh++) {
        const int j = 2 * jp + hh;
        float dg[4], du[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float gf = bf2f(g4[j][e]), uf = bf2f(u4[j][e]), df = bf2f(f2bf(acc[i][j][e] * p.alpha));
          const float sg = sigmoid_fast(gf);
          dg[e] = df * uf * sg * (1.0f + gf * (1.0f - sg));
          du[e] = df * (gf * sg);
        }
        pg[hh][0] = pack2bf(dg[0], dg[1]); pg[hh][1] = pack2bf(dg[2], dg[3]);
        pu[hh][0] = pack2bf(du[0], du[1]); pu[hh][1] = pack2bf(du[2], du[3]);
      }
      swap_rows16(pg[0][0], pg[1][0]);
      swap_rows16(pg[0][1], pg[1][1]);
      swap_rows16(pu[0][0], pu[1][0]);
      swap_rows16(pu[0][1], pu[1][1]);
      bf16_t* c = crow + jp * 32;
      const int nn = nbase + (2 * jp + (fg & 1)) * 16 + (fg >> 1) * 8;
      if (INNER || nn + 8 <= p.N) {
        *reinterpret_cast<uint4*>(c) = make_uint4(pg[0][0], pg[0][1], pg[1][0], pg[1][1]);
        *reinterpret_cast<uint4*>(c + p.N) = make_uint4(pu[0][0], pu[0][1], pu[1][0], pu[1][1]);
      } else if (nn < p.N) {   // N % 4 == 0 is enforced by the launcher
        uint2 o2;
        o2.x = pg[0][0]; o2.y = pg[0][1];
        *reinterpret_cast<uint2*>(c) = o2;
        o2.x = pu[0][0]; o2.y = pu[0][1];
        *reinterpret_cast<uint2*>(c + p.N) = o2;
      }
    }
  }
}

// SwiGLU forward fused into the gate|up product (act 4, 4-wave kernel only).  The weight rows are stored INTERLEAVED in blocks of 64
// (rows [128 b, 128 b + 64) = gate rows [64 b, 64 b + 64), rows [128 b + 64, 128 b + 128) = the matching up rows), so that the 128
// output columns of one wave are one block: its left 64-column quadrant holds gate, its right quadrant the SAME columns of up, in
// the same lanes and registers.  Writes the product itself ([gate64 | up64] blocks: the backward's stash, C) and
// h = silu(gate) * up (C2, [M, N / 2]) -- the stand-alone pass (slam_swiglu_fwd: 675 MB re-read + 338 MB written per Llama-3-8B
// layer at M = 11 780) disappears.  Same arithmetic as swiglu_fwd_kernel (elementwise.hip): on the bf16-ROUNDED gate / up.
// N is a multiple of 128 (launcher), so a wave's block is either wholly inside [0, N) or wholly outside; rows are guarded.
template <bool INNER>
__device__ __forceinline__ void gemm_epilogue_swiglu_fwd(const GemmParams& p, f32x4_t (&ag)[4][4], f32x4_t (&au)[4][4], int mbase,
                                                         int nbase, int frow, int fg) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int m = mbase + i * 16 + frow;
    if (!INNER && m >= p.M) continue;   // (both lanes of an exchanging pair share frow, hence m)
    bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + nbase + (fg & 1) * 16 + (fg >> 1) * 8;
    bf16_t* hrow = reinterpret_cast<bf16_t*>(p.C2) + (int64_t)m * p.ldc2 + (nbase >> 1) + (fg & 1) * 16 + (fg >> 1) * 8;
#pragma unroll
    for (int jp = 0; jp < 2; jp++) {
      unsigned pg[2][2], pu[2][2], ph[2][2];
#pragma unroll
      for (int hh = 0; hh < 2; hh++) {
        const int j = 2 * jp + hh;
        pg[hh][0] = pack2bf(ag[i][j][0], ag[i][j][1]);
        pg[hh][1] = pack2bf(ag[i][j][2], ag[i][j][3]);
        pu[hh][0] = pack2bf(au[i][j][0], au[i][j][1]);
        pu[hh][1] = pack2bf(au[i][j][2], au[i][j][3]);
        float hv[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const unsigned gw = pg[hh][e >> 1], uw = pu[hh][e >> 1];
          const float gf = __uint_as_float((e & 1) ? (gw & 0xFFFF0000u) : (gw << 16));
          const float uf = __uint_as_float((e & 1) ? (uw & 0xFFFF0000u) : (uw << 16));
          hv[e] = gf * sigmoid_fast(gf) * uf;
        }
        ph[hh][0] = pack2bf(hv[0], hv[1]);
        ph[hh][1] = pack2bf(hv[2], hv[3]);
      }
      swap_rows16(pg[0][0], pg[1][0]);
      swap_rows16(pg[0][1], pg[1][1]);
      swap_rows16(pu[0][0], pu[1][0]);
      swap_rows16(pu[0][1], pu[1][1]);
      swap_rows16(ph[0][0], ph[1][0]);
      swap_rows16(ph[0][1], ph[1][1]);
      *reinterpret_cast<uint4*>(crow + jp * 32) = make_uint4(pg[0][0], pg[0][1], pg[1][0], pg[1][1]);
      *reinterpret_cast<uint4*>(crow + 64 + jp * 32) = make_uint4(pu[0][0], pu[0][1], pu[1][0], pu[1][1]);
      *reinterpret_cast<uint4*>(hrow + jp * 32) = make_uint4(ph[0][0], ph[0][1], ph[1][0], ph[1][1]);
    }
  }
}

// fp32 outputs without bias / activation / residual (weight-gradient products, optionally accumulating): 16 bytes per lane as is
template <int FM, int FN, int WTM, int WTN, bool ACCUM>
__device__ __forceinline__ void gemm_epilogue_f32(const GemmParams& p, f32x4_t (&acc)[FM][FN], int m0, int n0, int wm, int wn,
                                                  int frow, int fg) {
#pragma unroll
  for (int i = 0; i < FM; i++) {
    const int m = m0 + wm * WTM + i * 16 + frow;
    if (m >= p.M) continue;
    float* crow = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n0 + wn * WTN + fg * 4;
#pragma unroll
    for (int j = 0; j < FN; j++) {
      if (n0 + wn * WTN + j * 16 + fg * 4 >= p.N) continue;
      float4 v = make_float4(acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha);
      if constexpr (ACCUM) {
        const float4 o = *reinterpret_cast<const float4*>(crow + j * 16);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      *reinterpret_cast<float4*>(crow + j * 16) = v;
    }
  }
}

template <int FM, int FN, int WTM, int WTN>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4_t (&acc)[FM][FN], int m0, int n0, int wm,
                                              int wn, int frow, int fg) {
  if (p.out_f32 && !p.bias && !p.res && p.act == 0) {
    if (p.accumulate) gemm_epilogue_f32<FM, FN, WTM, WTN, true>(p, acc, m0, n0, wm, wn, frow, fg);
    else gemm_epilogue_f32<FM, FN, WTM, WTN, false>(p, acc, m0, n0, wm, wn, frow, fg);
    return;
  }
  if (p.act == 3 && !p.out_f32 && !p.accumulate) {   // (the launcher guarantees no bias and a [gate | up] residual)
    if ((m0 + wm * WTM + WTM <= p.M) && (n0 + wn * WTN + WTN <= p.N))
      gemm_epilogue_swiglu_bwd<FM, FN, WTM, WTN, true>(p, acc, m0, n0, wm, wn, frow, fg);
    else
      gemm_epilogue_swiglu_bwd<FM, FN, WTM, WTN, false>(p, acc, m0, n0, wm, wn, frow, fg);
    return;
  }
  if (p.out_f32 || p.accumulate || p.act == 3 || p.alpha != 1.0f) {
    gemm_epilogue_generic<FM, FN, WTM, WTN>(p, acc, m0, n0, wm, wn, frow, fg);
    return;
  }
  const int key = (p.bias ? 1 : 0) | (p.res ? 2 : 0) | (p.act << 2);   // wave uniform
  // (tile extents: every caller's workgroup tile is WM x WN wave tiles of WTM x WTN; m0 / n0 may carry a quadrant offset, which
  // only makes this test conservative by less than a tile)
  const bool inner = (m0 + wm * WTM + WTM <= p.M) && (n0 + wn * WTN + WTN <= p.N);
#define SLAM_EPI(B_, A_, R_)                                                                                       \
  do {                                                                                                             \
    if (inner) gemm_epilogue_bf16<FM, FN, WTM, WTN, B_, A_, R_, true>(p, acc, m0, n0, wm, wn, frow, fg);          \
    else gemm_epilogue_bf16<FM, FN, WTM, WTN, B_, A_, R_, false>(p, acc, m0, n0, wm, wn, frow, fg);               \
  } while (0)
  switch (key) {
    case 0: SLAM_EPI(false, 0, false); break;
    case 1: SLAM_EPI(true, 0, false); break;
    case 2: SLAM_EPI(false, 0, true); break;
    case 3: SLAM_EPI(true, 0, true); break;
    case 4: SLAM_EPI(false, 1, false); break;
    case 5: SLAM_EPI(true, 1, false); break;
    case 6: SLAM_EPI(false, 1, true); break;
    case 7: SLAM_EPI(true, 1, true); break;
    case 8: SLAM_EPI(false, 2, false); break;
    case 9: SLAM_EPI(true, 2, false); break;
    case 10: SLAM_EPI(false, 2, true); break;
    default: SLAM_EPI(true, 2, true); break;
  }
#undef SLAM_EPI
}


// ---- hand-ordered loops (gemm_nt_w4_kernel): every instruction of the k-loop is an `asm volatile` statement ----
template <int I, int N, class F>
__device__ __forceinline__ void gemm_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    gemm_static_for<I + 1, N>(f);
  }
}
__device__ __forceinline__ void w4_mfma(f32x4_t& c, const bf16x8_t& a, const bf16x8_t& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <int OFF>
__device__ __forceinline__ void w4_lds_read(bf16x8_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// (M0 is not saved / restored: inside these all-asm loops nothing else uses it, and every issue slot counts)
__device__ __forceinline__ void w4_dma(__amdgpu_buffer_rsrc_t srd, unsigned voff, unsigned soff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :
               : "v"(voff), "s"(srd), "s"(soff), "s"(lds_dst)
               : "memory", "m0");
}

__device__ __forceinline__ void w4_gload(bf16x8_t& dst, __amdgpu_buffer_rsrc_t srd, unsigned voff, unsigned soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
template <int OFF>
__device__ __forceinline__ void w4_lds_write(unsigned addr, const bf16x8_t& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void w4_vmwait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

