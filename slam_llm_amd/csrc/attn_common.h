// Shared by the attention code (attention.hip; round 5 had a second translation unit): launch parameters, the workgroup -> (block, head,
// batch) numbering, MFMA / LDS / LDS-DMA helpers.  Everything lives in an anonymous namespace (one copy per translation unit); AttnParams
// crosses the boundary between them only as an opaque pointer (same header, same layout).
#pragma once
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace {

typedef u16x8_t frag_t;

struct AttnParams {
  const bf16_t* Q; int64_t ldq;     // [B*T, ldq], head h at column h*D
  const bf16_t* K; int64_t ldk;     // [B*T, ldk], kv head at column hk*D
  const bf16_t* V; int64_t ldv;     // row-major V (backward only)
  const bf16_t* Vt;                 // [B, Hkv, D, Tp]
  const bf16_t* Kt;                 // [B, Hkv, D, Tp] (backward dQ)
  const bf16_t* Qt;                 // [B, Hq, D, Tp]  (backward dK/dV)
  const bf16_t* dOt;                // [B, Hq, D, Tp]  (backward dK/dV)
  bf16_t* O; int64_t ldo;           // [B*T, ldo]
  const bf16_t* dO; int64_t lddo;   // [B*T, lddo]
  bf16_t* dQ; int64_t lddq;
  bf16_t* dK; int64_t lddk;
  bf16_t* dV; int64_t lddv;
  float* LSE;                       // [B, Hq, Tp] natural-log units
  float* Delta;                     // [B, Hq, Tp]
  const uint8_t* kmask;             // [B, Tp] 1 = attend (zero padded) or null
  int Tq, Tk, Tqp, Tkp, Hq, Hkv;  // query / key lengths and their 64-padded strides (Tq == Tk for self-attention)
  float scale;                      // softmax scale (1/sqrt(D))
  int qpre;                         // forward only: 1 = Q arrives multiplied by scale * log2(e) (slam_attn_fwd with a negative scale: the caller folded
                                    // the factor into a frozen query projection): the scores of the first product already are in the exponent's log2 units
  const float* rope_cos;            // backward only, nullable: [T, D/2] RoPE tables; when set dQ and dK are rotated back
  const float* rope_sin;            //   (d/dx of the forward rotation, position = row index) before they are stored
  const int* rope_pos;              // nullable [B*T]: explicit rotary position per row (packed / varlen batches)
  // packed ("varlen") self-attention: several sequences concatenated along T (B = 1).  seg_lo[q] = first key row query q
  // may attend (start of its sequence), seg_hi[k] = one past the last query row that may see key k (end of its
  // sequence); both non-decreasing, null for ordinary batches.  Causal only.
  const int* seg_lo;
  const int* seg_hi;
  // gated relative position bias (WavLM, forward only): score(q, k) = scale * q.k + rp_gate[b][h][q] * rp_tab[h][k - q + rp_T - 1]
  // (src/slam_llm/models/wavlm/modules.py:504-533: position_bias from the bucketed embedding of layer 0, gated per query);
  // rp_tab rows have stride rp_ld and 64 readable floats of slack before index 0 and after index 2 rp_T - 2
  const float* rp_gate;
  const float* rp_tab;
  int rp_T, rp_ld;
  // backward of that bias (unfrozen WavLM): the dQ kernel also writes dL/d(score) [B, Hq, Tq, Tkp] f32 (zeros where masked); the
  // gradients of the gate and of the table are row / diagonal reductions of it (attn_relpos_grad_kernels), no atomics
  float* rp_ds;
  // dropout on the attention probabilities (HF Blip2QFormer `attention_probs_dropout_prob`, train mode; D = 64 bidirectional
  // kernels only): P is normalised with the full row sum, then element (b, h, q, k) is kept with the counter-based mask of
  // slam_dropout_bf16 at index ((b*Hq + h)*Tqp + q)*Tkp + k and scaled by 1/(1-p) before the second product; the backward
  // kernels recompute the same mask.  drop_thresh = 0: none.
  unsigned drop_thresh;
  float drop_scale;
  unsigned long long drop_seed;
  const unsigned long long* drop_salt;   // null, or the device word of slam_set_dropout_salt (XORed into drop_seed: captured steps)
  // launch geometry (set by attn_launch): the logical grid is (gx sequence blocks, gy heads, gz batches), launched 1-D
  int gx, gy, gz;
  int xcd;   // 1: undo the hardware's round-robin workgroup -> XCD placement (attn_blk)
  int heavy; // causal, unpacked launches: +1 = sequence block 0 is the heaviest (dK / dV: key block 0 meets every query), -1 = the last one is
             // (forward, dQ: the last query block meets every key); attn_blk starts each XCD's heaviest blocks first.  0 = plain order
};

// Workgroup -> (sequence block, head, batch).  The hardware deals consecutive workgroups round-robin over the 8 XCDs (workgroup L
// runs on XCD L % 8), each with its own L2: with the plain (block, head, batch) grid the query blocks of one (batch, head) -- and the
// q-heads of a GQA group, which share one K/V -- land on all eight XCDs and every L2 fetches the same K/V (round 2 counters: 2.3-5.9x
// the algorithmic bytes, 4.3-4.9 TB/s of fabric traffic).  The grid is therefore launched 1-D and re-numbered with the bijection
// of the GEMM (gemm_bf16.hip): XCD x owns the contiguous run of logical ids [base(x), base(x+1)), and logical ids run sequence
// block fastest, then head (GQA siblings adjacent), then batch -- everything that shares a K/V (forward, dQ) or a Q/dO (dK/dV)
// tile set is co-resident on ONE XCD.
struct AttnBlk { int x, y, z; };
__device__ __forceinline__ AttnBlk attn_blk(const AttnParams& p) {
  int bid = blockIdx.x;
  if (p.xcd) {
    const int n = gridDim.x, xcd = bid & 7, q = n >> 3, r = n & 7;
    const int a = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;   // this XCD's run of logical ids: [a, a + cnt)
    int i = bid >> 3;
    bid = a + i;
    if (p.heavy != 0 && p.gx > 1) {
      // Round 5: longest block first.  Under a causal mask the sequence blocks of one (batch, head) differ 3 : 2 : 1 in work at T = 380
      // (12 : 8 : 4 tiles), the hardware starts an XCD's workgroups in id order and ids run sequence block fastest, so light and heavy
      // blocks alternated and whichever heavy block happened to start last set the launch's tail (744 dK / dV workgroups on 256 CUs:
      // ~3 waves of workgroups, the last one up to half a launch long).  The i-th workgroup an XCD starts now takes the i-th id of its
      // run in the order (heaviest sequence block first, ascending id inside a class): same ids, same XCD (L2 reuse across the blocks
      // of a head is what the run is for), every workgroup computes what it computed before -- results bit-identical.
      const int cnt = (xcd < r) ? q + 1 : q;
      const int gx = p.gx;
      for (int j = 0; j < gx; j++) {
        const int sb = p.heavy > 0 ? j : gx - 1 - j;
        int f = (sb - a) % gx;             // first id >= a with id % gx == sb is a + f
        if (f < 0) f += gx;
        const int c = f < cnt ? (cnt - f - 1) / gx + 1 : 0;
        if (i < c) {
          bid = a + f + i * gx;
          break;
        }
        i -= c;
      }
    }
  }
  AttnBlk o;
  o.x = bid % p.gx;
  const int t = bid / p.gx;
  o.y = t % p.gy;
  o.z = t / p.gy;
  return o;
}

// keep bits (bit r) of keys kb .. kb+3 (kb % 4 == 0) for query q of flattened (batch, head) bh
__device__ __forceinline__ unsigned attn_keep4(const AttnParams& p, int bh, int q, int kb) {
  const unsigned long long idx = ((unsigned long long)bh * (unsigned)p.Tqp + (unsigned)q) * (unsigned)p.Tkp + (unsigned)kb;
  const unsigned long long h64 = slam_mix64(slam_salted(p.drop_seed, p.drop_salt) ^ ((idx >> 2) * 0xD1342543DE82EF95ull));
  unsigned bits = 0;
#pragma unroll
  for (int e = 0; e < 4; e++) bits |= ((unsigned)((h64 >> (16 * e)) & 0xFFFFull) >= p.drop_thresh ? 1u : 0u) << e;
  return bits;
}

// gradient of HF's rotate_half RoPE for one row held as DF fragments of 4 consecutive head-dim elements per lane:
// dx1 = dy1 cos + dy2 sin, dx2 = dy2 cos - dy1 sin with (1, 2) = (d, d + D/2) -> fragments (df, df + DF/2) of the same lane
template <int DF>
__device__ __forceinline__ void rope_grad_inplace(f32x4_t (&v)[DF], const float* cosT, const float* sinT, int pos, int D,
                                                  int g) {
#pragma unroll
  for (int df = 0; df < DF / 2; df++) {
    const float4 c4 = *reinterpret_cast<const float4*>(cosT + (int64_t)pos * (D / 2) + df * 16 + 4 * g);
    const float4 s4 = *reinterpret_cast<const float4*>(sinT + (int64_t)pos * (D / 2) + df * 16 + 4 * g);
    const float cs[4] = {c4.x, c4.y, c4.z, c4.w}, sn[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float a = v[df][r], b = v[df + DF / 2][r];
      v[df][r] = a * cs[r] + b * sn[r];
      v[df + DF / 2][r] = b * cs[r] - a * sn[r];
    }
  }
}

__device__ __forceinline__ f32x4_t mfma16(frag_t a, frag_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                 __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ frag_t zero_frag() {
  frag_t z;
#pragma unroll
  for (int e = 0; e < 8; e++) z[e] = 0;
  return z;
}
__device__ __forceinline__ frag_t pack_frag(f32x4_t x, f32x4_t y) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  u32x4_t w;
  w[0] = pack2bf(x[0], x[1]);
  w[1] = pack2bf(x[2], x[3]);
  w[2] = pack2bf(y[0], y[1]);
  w[3] = pack2bf(y[2], y[3]);
  return __builtin_bit_cast(frag_t, w);
}
// v_exp_f32 directly: arguments here are <= 0 (or -inf), no denormal-range fix-up needed
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ frag_t join_frag(u16x4_t lo, u16x4_t hi) {
  frag_t f;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    f[e] = lo[e];
    f[4 + e] = hi[e];
  }
  return f;
}
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ unsigned lds_offset_of(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

template <bool FIRST, class A, class B>
__device__ __forceinline__ auto& pick_ref(A& a, B& b) {
  if constexpr (FIRST) return a;
  else return b;
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// hand-placed LDS reads: the result register is "ready" for the compiler at once, so every use MUST sit behind an lds_wait
// that names it (LDS operations return in order: lgkmcnt(N) = all but the youngest N have landed)
template <int OFF>
__device__ __forceinline__ frag_t lds_read128(unsigned addr) {
  frag_t r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ u32x2_t lds_read64(unsigned addr) {
  u32x2_t r;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ void lds_landed(frag_t& r) { asm volatile("" : "+v"(r)); }
template <int N, class... T>
__device__ __forceinline__ void lds_wait(T&... regs) {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  (lds_landed(regs), ...);
}
// reductions over the four 16-lane groups (lanes l, l^16, l^32, l^48) with gfx950's v_permlane16_swap / v_permlane32_swap: of the
// two results one is the lane's own value and the other its partner's, so max / sum need no select -- and, unlike __shfl_xor
// (ds_bpermute), they do not go through the LDS queue, whose counter the hand-placed reads of the tile loop are counting on
// (inline asm: with this hipcc the builtins' second result folds to the first -- `r[1]` of __builtin_amdgcn_permlane16_swap
// compiles to `extractvalue 0` -- and the two operands must be different registers, which "+v" twice guarantees)
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float max_across_groups(float v) {
  float a = v, b = v;
  swap16(a, b);
  a = b = fmaxf(a, b);
  swap32(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float sum_across_groups(float v) {
  float a = v, b = v;
  swap16(a, b);
  a = b = a + b;
  swap32(a, b);
  return a + b;
}

// XOR key of the 16-byte chunks of K-tile row `row` (forward kernel): a fragment reads rows 8(i/4) + i%4 (+ 4f' + 32a), i = 0..15;
// the key is distinct over them per 256-byte bank window (D = 128: one row per window, 16 keys; D = 64: two rows, 8 keys)
template <int D>
__device__ __forceinline__ int fwd_swz(int row) {
  const int i = (((row >> 3) & 3) << 2) | (row & 3);
  return D == 128 ? i : (i >> 1);
}

// ------------------------------------------------------------------------------------------
// The DMA is issued from inline asm: hipcc (ROCm 7.2) protects every LDS read that follows a global_load_lds BUILTIN with
// s_waitcnt vmcnt(0) (it cannot tell the ring stages apart), which drains the three tiles in flight on every iteration.  An
// asm DMA is invisible to that bookkeeping; the counted vmcnt + barrier below order it by hand.  M0 (the DMA's LDS base) is
// compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void glds16_asm(const void* g, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(g), "s"(lds_dst_uniform)
               : "memory");
}
// buffer-descriptor form (buffer_load_dwordx4 ... offen lds): one 32-bit byte offset per lane, range-checked against
// num_records (rows past the end of the tensor read as zeros: no clamps), the tile part of the address is a scalar
__device__ __forceinline__ void bufdma16_asm(__amdgpu_buffer_rsrc_t srd, unsigned voff, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(srd), "s"(lds_dst_uniform)
               : "memory");
}

// ------------------------------------------------------------------------------------------
// Transposed MFMA operands straight from the ROW-MAJOR tiles (round 4): ds_read_b64_tr_b16.  Inside every 16-lane group, result lane i,
// element j = element (i & 3) of the 8 bytes addressed by lane 4 j + (i >> 2) (tools/probes/tr_probe.py checks this on random addresses).
// With lane i pointing at row 8 g + 4 half + (i >> 2), columns 16 df + 4 (i & 3) .. + 3 of a row-major [rows][D] tile, lane i receives
// rows 8 g + 4 half .. + 3 of COLUMN 16 df + i: two such reads (half = 0, 1) are the A operand "row = head-dim element 16 df + li,
// k-slots = tile rows 8 g .. 8 g + 7" that the second products of all three kernels (O^T += V^T P^T, dQ^T += K^T dS^T, dV^T += dO^T P,
// dK^T += Q^T dS) used to read from transposed copies ([B, H, D, Tp] in HBM, written by slam_head_rope_transpose: three launches per
// layer, and half of the backward kernels' LDS-DMA pieces).
// Swizzle: the 16-byte chunks of tile row `row` are XOR-ed with tr_swz<D>(row); the same key keeps the b128 fragment reads of the FIRST
// products (16 rows 8 (i / 4) + 4 f + i % 4 per fragment) and these reads (8 rows x 32 bytes per half-wave) free of bank conflicts
// (measured: 333 cycles per 16 reads and wave = the conflict-free floor; the round-2/3 keys cost 512, no swizzle 1707).
template <int D>
__device__ __forceinline__ int tr_swz(int row) {
  return D == 128 ? (((row & 3) << 1) | (((row >> 3) & 1) << 3)) : ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2) | (row & 1));
}
template <int OFF>
__device__ __forceinline__ u32x2_t lds_read_tr(unsigned addr) {
  u32x2_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
struct TrFrag {   // the two halves of one transposed A operand (k-slots 0-3 | 4-7 of the lane's group)
  u32x2_t lo, hi;
};
__device__ __forceinline__ frag_t tr_join(const TrFrag& t) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  u32x4_t w;
  w[0] = t.lo[0]; w[1] = t.lo[1]; w[2] = t.hi[0]; w[3] = t.hi[1];
  return __builtin_bit_cast(frag_t, w);
}
__device__ __forceinline__ void lds_landed(TrFrag& r) { asm volatile("" : "+v"(r.lo), "+v"(r.hi)); }
// per-lane byte offset (inside a row-major [rows][D] tile) of the first transposed read: row 8 g + (li >> 2), the 8-byte half li & 1 of
// logical chunk (li & 3) >> 1; df enters as XOR (df << 5) (it shares the chunk bits with the key), half / tile row blocks as immediates
template <int D>
__device__ __forceinline__ unsigned tr_lane_off(int g, int li) {
  const int row = 8 * g + (li >> 2);
  return (unsigned)(row * (D * 2) + (((((li & 3) >> 1)) ^ tr_swz<D>(row)) << 4) + (li & 1) * 8);
}


}  // namespace
