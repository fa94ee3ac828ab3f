// Flash-style attention on MFMA for gfx950: encoder (bidirectional, Whisper blocks reached from
// src/slam_llm/models/encoder.py:26-27) and LLM (causal + key-padding, GQA; HF LlamaAttention with the
// 4-D causal^padding mask, transformers/models/llama/modeling_llama.py:217-282, called through
// src/slam_llm/models/slam_model.py:400), forward and backward.
//
// Layout trick (no P transposes, no LDS round trip for P):
//   v_mfma_f32_16x16x32_bf16 computes D[i][j] = sum_k A[i][k] * B[k][j]; lane l supplies A[l&15][8*(l>>4)+e]
//   and B[8*(l>>4)+e][l&15] (e = 0..7) and receives D[4*(l>>4)+r][l&15] (r = 0..3).  The k index only has
//   to be enumerated identically for A and B, so:
//     S^T = K . Q^T          (A = K rows, B = Q rows)        -> lane owns query (l&15), 4 consecutive keys
//     O^T = V^T . P^T        (A = V^T rows = head-dim, B = P) -> lane owns the same query, 4 consecutive d
//   Two S^T fragments (32 keys) already ARE a B operand for the second product, provided the V^T operand
//   enumerates keys in the same order.  The forward and the ring dK/dV kernel feed the first product with PERMUTED rows
//   (fragment row i = tile row 8(i/4) + 4f + i%4), which makes that order "k-slot (g, e) = key 32a + 8g + e": eight
//   contiguous keys, one 16-byte read of the transposed operand; the round-1 backward kernels use natural rows and
//   k-slot (g, e) = key 16*(2a + e/4) + 4g + e%4 (two 8-byte reads).  V (and, for the backward, K, Q and dO) are
//   therefore also kept transposed ([B,H,D,Tp], written by slam_head_rope_transpose).
//   All per-query softmax state (m, l, LSE, Delta) is lane-local; row reductions are two xor-shuffles.
//
// Shipped forms (round 2): attn_fwd_kernel<..., DMA = true>, attn_bwd_dq_ring_kernel, attn_bwd_dkdv_ring_kernel -- operand tiles by
// descriptor LDS-DMA into rings of 2-4 stages, counted vmcnt / lgkmcnt waits, one raw barrier per tile, LDS reads placed by hand.
// The register-staged kernels (attn_fwd_kernel<..., DMA = false>, attn_bwd_dq_kernel, attn_bwd_dkdv_kernel) remain for tensors
// beyond the descriptors' 32-bit offsets, <= 64 queries, attention dropout, and as A/B references in tools/.
//
// Masking follows HF: key j is visible to query i iff j <= i (causal) and key_mask[b][j]; query rows are
// never masked.  A row with no visible key yields O = 0, LSE = +inf (P = 0 in the backward)
// so pad rows stay finite.  Positions/rows beyond T are handled by clamped/zero loads and guarded stores.
#include "attn_common.h"

namespace {

__device__ unsigned long long g_attn_probe[2 * 16 * 8];   // tools: cycle stamps of two waves of one workgroup (PROBE forms)

// forward: 4 waves x 32 query rows per workgroup, 64-key K / V^T tiles staged through LDS
// ------------------------------------------------------------------------------------------
// QF = 16-row query fragments per wave (2: 128-query workgroups; 1: 64-query workgroups, fewer VGPRs -> more waves per SIMD)
// TRV (round 4, shipped): the V tile is staged ROW-MAJOR ([64 keys][D], straight from the fused QKV buffer: no [B,H,D,Tp] copy of V) and
// the V^T operand of O^T += V^T P^T is read with ds_read_b64_tr_b16 (two 8-byte reads per fragment instead of one 16-byte read)
// QS (round 5; LSE-less launches of the PLAIN form with p.qpre: the frozen Whisper encoder, whose query projection carries
// scale * log2(e) -- folded into W_q, b_q in fp32 when the checkpoint is loaded, so Q is rounded to bf16 once, like the unscaled Q of the
// reference path): every tile behind the first starts the S^T accumulators at -m (the running maximum): the first product then delivers
// s * scale * log2(e) - m and P = exp2 of it -- no v_fma per score in the softmax.
template <int D, bool CAUSAL, int QF, bool RP = false, bool DROP = false, bool PROBE = false, bool DMA = false, bool PLAIN = false, bool TRV = false, bool QS = false>
__global__ __launch_bounds__(256, (PLAIN ? 3 : 1)) void attn_fwd_kernel(AttnParams p) {
  static_assert(!QS || (PLAIN && !CAUSAL && !RP && !DROP), "QS is a variant of the mask-free bidirectional form");
  constexpr int KD = D / 32;
  constexpr int DF = D / 16;
  constexpr int KROWB = D * 2;
  constexpr int KCH = D / 8;       // 16-byte chunks per K row
  constexpr int KCM = KCH - 1;
  constexpr int KI = 64 * KCH / 256;
  constexpr int VI = D * 8 / 256;
  // DMA form: the K / V^T tiles go HBM -> LDS by descriptor LDS-DMA into a ring of NS stages (counted vmcnt, one raw barrier
  // per tile) instead of through staging registers, two __syncthreads and 4-8 ds_write_b128 per thread and tile
  constexpr int NS = DMA ? (D == 128 ? 2 : 3) : 1;
  constexpr int STG = 64 * KROWB + D * 128;
  constexpr int NPW = STG / 1024 / 4;   // 1 KiB DMA pieces per wave and tile (K first, then V^T)
  __shared__ __attribute__((aligned(16))) char lds[NS * STG];
  char* ldsK = lds;
  char* ldsV = lds + 64 * KROWB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = DMA ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const AttnBlk blk = attn_blk(p);
  const int b = blk.z, h = blk.y;
  const int hk = h / (p.Hq / p.Hkv);
  const int Tq = p.Tq, Tk = p.Tk, Tqp = p.Tqp, Tkp = p.Tkp;
  // PLAIN: no key mask, no packed segments (the Whisper encoder of a padded batch): the pointers are compile-time null, the
  // bookkeeping of those paths leaves the register budget, and with two key fragments (not four) requested ahead and the first V^T
  // reads issued behind the softmax the kernel fits 168 VGPRs = three waves per SIMD
  const uint8_t* kmask_ = PLAIN ? nullptr : p.kmask;
  const int* seg_lo_ = PLAIN ? nullptr : p.seg_lo;
  const int* seg_hi_ = PLAIN ? nullptr : p.seg_hi;
  constexpr int QW = 16 * QF;   // query rows per wave
  const int qb0 = blk.x * (4 * QW), qw0 = qb0 + wave * QW;

  const float sl2 = p.qpre ? 1.0f : p.scale * LOG2E;   // what turns a score of the first product into the exponent's log2 units
  const float ksc = QS ? 1.0f : sl2;
  frag_t qf[QF][KD];
#pragma unroll
  for (int f = 0; f < QF; f++) {
    const int q = qw0 + f * 16 + li;
#pragma unroll
    for (int kd = 0; kd < KD; kd++) {
      qf[f][kd] = (q < Tq) ? *reinterpret_cast<const frag_t*>(p.Q + ((int64_t)b * Tq + q) * p.ldq + h * D + kd * 32 + g * 8)
                          : zero_frag();
    }
  }
  f32x4_t o[QF][DF];
#pragma unroll
  for (int f = 0; f < QF; f++)
#pragma unroll
    for (int df = 0; df < DF; df++) o[f][df] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float mrow[QF], lrow[QF];
#pragma unroll
  for (int f = 0; f < QF; f++) {
    mrow[f] = -INFINITY;
    lrow[f] = 0.f;
  }

  // packed batches (seg_lo/seg_hi over QUERY rows, both non-decreasing): query q sees keys seg_lo[q] <= k < seg_hi[q]
  // (and k <= q when causal; the bidirectional form is the ragged Whisper / HuBERT encoder: one clip = one segment).
  // Keys before the start of the block's first sequence / past the end of its last one are never visible.
  const bool seg_both = !CAUSAL && seg_lo_ != nullptr;
  const int kend = CAUSAL ? min(Tk, qb0 + 4 * QW) : (seg_both ? min(Tk, seg_hi_[(int64_t)b * Tq + min(qb0 + 4 * QW - 1, Tq - 1)]) : Tk);
  const int ntiles = (kend + 63) / 64;
  const int tbeg = seg_lo_ ? seg_lo_[(int64_t)b * Tq + min(qb0, Tq - 1)] / 64 : 0;
  int qlo[QF], qhi[QF];
#pragma unroll
  for (int f = 0; f < QF; f++) {
    qlo[f] = 0;
    qhi[f] = 0x7fffffff;
  }
  int lo_wave_max = 0;            // largest sequence start among this wave's queries
  int hi_wave_min = 0x7fffffff;   // smallest sequence end among them
  if (seg_lo_) {
#pragma unroll
    for (int f = 0; f < QF; f++) qlo[f] = seg_lo_[(int64_t)b * Tq + min(qw0 + f * 16 + li, Tq - 1)];
    lo_wave_max = seg_lo_[(int64_t)b * Tq + min(qw0 + QW - 1, Tq - 1)];
    if (seg_both) {
#pragma unroll
      for (int f = 0; f < QF; f++) qhi[f] = seg_hi_[(int64_t)b * Tq + min(qw0 + f * 16 + li, Tq - 1)];
      hi_wave_min = seg_hi_[(int64_t)b * Tq + min(qw0, Tq - 1)];
    }
  }

  // key-padding bitmap (wave-uniform): bit t = the 64-key tile t holds a padded key.  Tiles without one take the mask-free
  // path although a mask was passed (an LLM batch is padded at one end: most tiles of most rows are clean).
  unsigned long long padtiles = 0;
  bool pad_known = true;
  if (kmask_) {
    const int ndw = Tkp >> 2;
    if (ndw > 64 * 16) {
      pad_known = false;   // more than 64 tiles: every tile takes the masked path
    } else {
      const unsigned* m32 = reinterpret_cast<const unsigned*>(kmask_ + (int64_t)b * Tkp);
      for (int c = 0; c * 64 < ndw; c++) {
        const int dw = c * 64 + lane;
        const unsigned v = dw < ndw ? m32[dw] : 0x01010101u;
        const unsigned long long bal = __ballot(((v - 0x01010101u) & ~v & 0x80808080u) != 0);   // some byte of v is zero
        const unsigned nib = ((bal & 0xffffull) ? 1u : 0u) | ((bal & 0xffff0000ull) ? 2u : 0u) | (((bal >> 32) & 0xffffull) ? 4u : 0u) |
                             ((bal >> 48) ? 8u : 0u);
        padtiles |= (unsigned long long)nib << (4 * c);
      }
    }
  }

  float rp_g[QF];
#pragma unroll
  for (int f = 0; f < QF; f++)
    rp_g[f] = RP ? p.rp_gate[((int64_t)b * p.Hq + h) * Tqp + min(qw0 + f * 16 + li, Tq - 1)] * LOG2E : 0.f;

  frag_t kreg[KI], vreg[VI];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < KI; i++) {
      const int item = tid + i * 256;
      const int row = item / KCH, c = item % KCH;
      const int key = k0 + row;
      kreg[i] = (key < Tk) ? *reinterpret_cast<const frag_t*>(p.K + ((int64_t)b * Tk + key) * p.ldk + hk * D + c * 8)
                          : zero_frag();
    }
#pragma unroll
    for (int i = 0; i < VI; i++) {
      const int item = tid + i * 256;
      if constexpr (TRV) {
        const int row = item / KCH, c = item % KCH;
        const int key = k0 + row;
        vreg[i] = (key < Tk) ? *reinterpret_cast<const frag_t*>(p.V + ((int64_t)b * Tk + key) * p.ldv + hk * D + c * 8) : zero_frag();
      } else {
        const int d = item >> 3, c = item & 7;
        vreg[i] = *reinterpret_cast<const frag_t*>(p.Vt + ((int64_t)(b * p.Hkv + hk) * D + d) * Tkp + k0 + c * 8);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < KI; i++) {
      const int item = tid + i * 256;
      const int row = item / KCH, c = item % KCH;
      *reinterpret_cast<frag_t*>(ldsK + row * KROWB + ((c ^ fwd_swz<D>(row)) << 4)) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < VI; i++) {
      const int item = tid + i * 256;
      if constexpr (TRV) {
        const int row = item / KCH, c = item % KCH;
        *reinterpret_cast<frag_t*>(ldsV + row * KROWB + ((c ^ tr_swz<D>(row)) << 4)) = vreg[i];
      } else {
        const int d = item >> 3, c = item & 7;
        *reinterpret_cast<frag_t*>(ldsV + d * 128 + ((c ^ ((d >> 1) & 7)) << 4)) = vreg[i];
      }
    }
  };

  const bool prb = PROBE && blk.x == 5 && blk.y == 7 && blk.z == 3 && (wave == 0 || wave == 3);
  auto stamp = [&](int it, int i) {
    if constexpr (PROBE) {
      if (prb) {
        const unsigned long long t = __builtin_readcyclecounter();
        if (lane == 0 && it < 16) g_attn_probe[((wave ? 1 : 0) * 16 + it) * 8 + i] = t;
      }
    }
  };
  // ---- DMA form: this wave owns pieces j = wave + 4 u (u < NPW) of the STG / 1024 that make a stage; the first half of them are
  // K rows (lane -> row, 16-byte chunk; the XOR swizzle is applied to the SOURCE chunk, the LDS image is lane-linear), the second
  // half V^T rows.  K rows past the end of the tensor read as zeros; rows past Tk of a batch in the middle read the next batch's
  // rows, which only tiles on the masked path (k0 + 64 > Tk) can contain. ----
  unsigned voff[NPW], dsto[NPW];
  __amdgpu_buffer_rsrc_t srd_k, srd_vt;
  unsigned lds0 = 0, ldk2 = 0, ldv2 = 0;
  if constexpr (DMA) {
    lds0 = __builtin_amdgcn_readfirstlane(lds_offset_of(lds));
    ldk2 = (unsigned)p.ldk * 2u;
    ldv2 = TRV ? (unsigned)p.ldv * 2u : 0u;
    const int nB = p.gz;
    srd_k = __builtin_amdgcn_make_buffer_rsrc((void*)p.K, 0, (unsigned)((((int64_t)nB * Tk - 1) * p.ldk + (int64_t)p.Hkv * D) * 2), 0x00020000);
    if constexpr (TRV) srd_vt = __builtin_amdgcn_make_buffer_rsrc((void*)p.V, 0, (unsigned)((((int64_t)nB * Tk - 1) * p.ldv + (int64_t)p.Hkv * D) * 2), 0x00020000);
    else srd_vt = __builtin_amdgcn_make_buffer_rsrc((void*)p.Vt, 0, (unsigned)((int64_t)nB * p.Hkv * D * Tkp * 2), 0x00020000);
#pragma unroll
    for (int u = 0; u < NPW; u++) {
      const int j = u * 4 + wave;                 // piece of the stage
      if (u < NPW / 2) {                          // K: [64][D] row-major
        const int row = j * (1024 / KROWB) + lane / KCH, c = lane % KCH;
        voff[u] = (unsigned)(((int64_t)b * Tk + row) * p.ldk + hk * D + ((c ^ fwd_swz<D>(row)) << 3)) * 2u;
      } else if constexpr (TRV) {                 // V: [64][D] row-major like K, swizzled for the transposed reads
        const int jv = j - NPW * 2;
        const int row = jv * (1024 / KROWB) + lane / KCH, c = lane % KCH;
        voff[u] = (unsigned)(((int64_t)b * Tk + row) * p.ldv + hk * D + ((c ^ tr_swz<D>(row)) << 3)) * 2u;
      } else {                                    // V^T: [D][64 keys], 128-byte rows
        const int jv = j - NPW * 2;               // piece inside the V^T sub-tile
        const int d = jv * 8 + (lane >> 3), c = lane & 7;
        voff[u] = (unsigned)((((int64_t)b * p.Hkv + hk) * D + d) * Tkp + ((c ^ ((d >> 1) & 7)) << 3)) * 2u;
      }
      dsto[u] = (unsigned)(j * 1024);
    }
    // every ordinary load above must have RETURNED before the first asm DMA is issued (see attn_bwd_dkdv_ring_kernel)
#pragma unroll
    for (int f = 0; f < QF; f++) {
#pragma unroll
      for (int kd = 0; kd < KD; kd++) asm volatile("" : "+v"(qf[f][kd]));
      asm volatile("" : "+v"(qlo[f]), "+v"(qhi[f]), "+v"(rp_g[f]));
    }
    asm volatile("" : "+s"(lo_wave_max), "+s"(hi_wave_min), "+s"(padtiles));
  }
  auto issue = [&](int tile, int stg) {
    if constexpr (DMA) {
      const int k0 = min(tile, ntiles - 1) * 64;   // past the end: the last tile again, into a stage nobody reads
      const unsigned st = lds0 + (unsigned)(stg * STG);
#pragma unroll
      for (int u = 0; u < NPW; u++) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(st + dsto[u]);
        const unsigned so = (u < NPW / 2) ? (unsigned)k0 * ldk2 : (TRV ? (unsigned)k0 * ldv2 : (unsigned)k0 * 2u);
        bufdma16_asm(u < NPW / 2 ? srd_k : srd_vt, voff[u] + __builtin_amdgcn_readfirstlane(so), dst);
      }
    }
  };
  int tb = tbeg, stage = 0;
  if constexpr (DMA) {
    asm volatile("" : "+s"(tb));
    if (ntiles > tb) {
      issue(tb, 0);
      if constexpr (NS == 3) issue(tb + 1, 1);
    }
  } else {
    gload(tbeg * 64);
  }
  for (int it = tb; it < ntiles; it++) {
    const int k0 = it * 64;
    stamp(it, 0);
    if constexpr (DMA) {
      // tile `it` has landed when at most the younger tile's DMA of this wave is outstanding; after the barrier every wave's
      // share has, and everybody is done reading the stage that the next DMA refills
      if constexpr (NS == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      stamp(it, 1);
      ldsK = lds + stage * STG;
      ldsV = ldsK + 64 * KROWB;
      const int nxt = stage + NS - 1 >= NS ? stage - 1 : stage + NS - 1;   // (stage + NS - 1) % NS
      stamp(it, 2);
      issue(it + NS - 1, nxt);
      stage = stage + 1 == NS ? 0 : stage + 1;
    } else {
      __syncthreads();
      stamp(it, 1);
      lstore();
      __syncthreads();
      stamp(it, 2);
      if (it + 1 < ntiles) gload(k0 + 64);
    }
    stamp(it, 3);
    if (CAUSAL && k0 > qw0 + QW - 1) continue;  // whole tile is in this wave's future

    // ---- S^T = K . Q^T ----
    // Key fragment kf = 2a + f' covers tile keys 32a + 8(i/4) + 4f' + i%4 (i = A-operand row): the eight P values a lane then
    // holds for the pair a are the CONTIGUOUS keys 32a + 8g .. 32a + 8g + 7, so the V^T operand of the second product is one
    // 16-byte LDS read (it was two 8-byte reads and a register shuffle).  LDS reads are asm with counted waits: all K fragments
    // are requested up front, products start as they land; the V^T reads of the first pair are requested before the softmax
    // arithmetic and land behind it.
    // QS (a PLAIN form: no mask, no segments): a tile behind the first one (every row has a finite maximum by then) starts its
    // accumulators at -m; the scores of keys past Tk of the last tile are replaced by -inf behind the product (P = 0 on every path: the QS form has no masked path)
    const bool pre = QS && it > tb;
    const bool tail = QS && k0 + 64 > Tk;
    f32x4_t s[QF][4];
    // FUSE (QS attempts): the exponentials (and row-sum adds) of key fragment kf - 1 are placed between the MFMAs of fragment kf -- a SIMD
    // runs its waves' MFMA and VALU phases one after the other unless they alternate inside one instruction stream (profiles/r05_attention_isa.md:
    // 512 cycles of MFMA + 512 of v_exp_f32 + 280 of other VALU per wave and tile = the measured tile time); sched_group_barrier pins the pattern
    float fz[QF];   // FUSE: the row sum per query fragment (one chain: its adds sit between MFMAs, their latency is covered)
    auto s_product = [&](const bool from_minus_m, auto tail_c, const int li, const int g, auto fuse_c) {   // (li, g: see the second call)
      constexpr bool TAIL = decltype(tail_c)::value;
      constexpr bool FUSE = decltype(fuse_c)::value;
      auto exp_pair = [&](auto kfe, int f, int r) {   // P of elements r, r + 1 of fragment kfe in place of their scores, row-sum chains updated
        const float pv0 = fast_exp2(s[f][kfe][r]);
        const float pv1 = fast_exp2(s[f][kfe][r + 1]);
        s[f][kfe][r] = pv0;
        s[f][kfe][r + 1] = pv1;
        fz[f] += pv0;
        fz[f] += pv1;
      };
      if constexpr (FUSE) {
#pragma unroll
        for (int f = 0; f < QF; f++) fz[f] = 0.f;
      }
      const int i4 = (li >> 2) * 8 + (li & 3);
      const unsigned kbase = lds_offset_of(ldsK) + (unsigned)(i4 * KROWB);
      unsigned ka[KD];
#pragma unroll
      for (int kd = 0; kd < KD; kd++) ka[kd] = kbase + (unsigned)(((kd * 4 + g) ^ fwd_swz<D>(i4)) << 4);   // (swz ignores bits 2, 5 of the row)
      frag_t kfr[4][KD];
      // AHEAD key fragments are requested before the first product; fragment kf + AHEAD is requested when kf's operands have
      // landed (D = 128: two ahead = 8 reads in flight -- all four up front cost 33 more VGPRs and measured 6 % slower)
      constexpr int AHEAD = (D == 128 || PLAIN) ? 2 : 4;
      static_for<0, AHEAD>([&](auto kf) {
        static_for<0, KD>([&](auto kd) { kfr[kf][kd] = lds_read128<((kf >> 1) * 32 + (kf & 1) * 4) * KROWB>(ka[kd]); });
      });
      static_for<0, 4>([&](auto kf) {
        constexpr int younger = (kf + AHEAD <= 4 ? AHEAD - 1 : 3 - kf) * KD;   // reads issued after fragment kf's
        static_for<0, KD>([&](auto kd) { lds_wait<younger + (KD - 1 - kd)>(kfr[kf][kd]); });
        if constexpr (kf + AHEAD < 4)
          static_for<0, KD>([&](auto kd) {
            kfr[kf + AHEAD][kd] = lds_read128<(((kf + AHEAD) >> 1) * 32 + ((kf + AHEAD) & 1) * 4) * KROWB>(ka[kd]);
          });
#pragma unroll
        for (int f = 0; f < QF; f++) {
          const float a0 = (QS && from_minus_m) ? -mrow[f] : 0.f;
          f32x4_t acc = f32x4_t{a0, a0, a0, a0};
#pragma unroll
          for (int kd = 0; kd < KD; kd++) {
            acc = mfma16(kfr[kf][kd], qf[f][kd], acc);
            if constexpr (FUSE && kf >= 1) {
              // behind every MFMA of fragment kf: one pair of exponentials (+ two adds) of fragment kf - 1 (QF * KD MFMAs, QF * 2 pairs),
              // fenced so that the order in the instruction stream IS this order
              static_assert(KD == 2, "one pair of fragment kf - 1 per MFMA of fragment kf");
              __builtin_amdgcn_sched_barrier(0);
              exp_pair(std::integral_constant<int, (kf >= 1 ? kf - 1 : 0)>{}, f, 2 * kd);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if constexpr (TAIL) {   // element r of fragment kf is key k0 + 32 (kf / 2) + 8 g + 4 (kf % 2) + r
            // keys past Tk: the score is DISCARDED (a select behind the product, last tile only), not "-inf + whatever the LDS rows past
            // the tile hold" -- those rows are the next batch item's K (or descriptor zeros), and an Inf / NaN there must stay that
            // item's problem (-inf + NaN = NaN reached this item's row sums)
#pragma unroll
            for (int r = 0; r < 4; r++)
              if (k0 + (kf >> 1) * 32 + 8 * g + (kf & 1) * 4 + r >= Tk) acc[r] = -INFINITY;
          }
          s[f][kf] = acc;
        }
      });
      if constexpr (FUSE) {
#pragma unroll
        for (int f = 0; f < QF; f++)
#pragma unroll
          for (int r = 0; r < 4; r += 2) exp_pair(std::integral_constant<int, 3>{}, f, r);
      }
    };
    if constexpr (QS) {
      if (pre) {
        if (tail) s_product(true, std::true_type{}, li, g, std::true_type{});
        else s_product(true, std::false_type{}, li, g, std::true_type{});
      } else if (tail) {
        s_product(false, std::true_type{}, li, g, std::false_type{});
      } else {
        s_product(false, std::false_type{}, li, g, std::false_type{});
      }
    } else {
      s_product(false, std::false_type{}, li, g, std::false_type{});
    }
    stamp(it, 4);
    const unsigned vbase = lds_offset_of(ldsV) + (unsigned)(li * 128);
    unsigned va[2];
#pragma unroll
    for (int a = 0; a < 2; a++) va[a] = vbase + (unsigned)(((a * 4 + g) ^ ((li >> 1) & 7)) << 4);   // rows d = df*16 + li: (d >> 1) & 7 = (li >> 1) & 7
    frag_t vfr[2][DF];
    // TRV: fragment t = a * DF + df of the V^T operand = two transposed 8-byte reads of V-tile rows 32 a + 8 g + 4 half + (li >> 2);
    // TPRE fragments are requested before the softmax (12 reads at most: the LDS counter holds 15), TAH ahead of their products after it
    constexpr int NT = 2 * DF;
    constexpr int TPRE = PLAIN ? 0 : (DF < 6 ? DF : 6);
    constexpr int TAH = DF < 6 ? DF : 6;
    TrFrag vtr[TRV ? NT : 1];
    unsigned vta[TRV ? DF : 1];
    if constexpr (TRV) {
      const unsigned t0 = lds_offset_of(ldsV) + tr_lane_off<D>(g, li);
#pragma unroll
      for (int df = 0; df < DF; df++) vta[df] = t0 ^ (unsigned)(df << 5);   // (the stage base is a multiple of 1 KiB)
      static_for<0, TPRE>([&](auto t) {
        vtr[t].lo = lds_read_tr<((t / DF) * 32) * KROWB>(vta[t % DF]);
        vtr[t].hi = lds_read_tr<((t / DF) * 32 + 4) * KROWB>(vta[t % DF]);
      });
    } else {
      if constexpr (!PLAIN) static_for<0, DF>([&](auto df) { vfr[0][df] = lds_read128<df * 16 * 128>(va[0]); });
    }
    // ---- online softmax (per query = lane&15, replicated over the 4 lane groups) ----
    // The running maximum only moves when a tile exceeds it by more than 2^8 (in the exponent's log2 units): P stays <= 256,
    // exact in fp32 / bf16, and the O / l rescale becomes a rare wave-uniform branch instead of 2 exp + 16 multiplies per tile
    // interior tiles (no padded key, fully inside [0, Tk), fully below the causal diagonal of this wave) skip all masking
    const bool tile_pad = kmask_ != nullptr && (!pad_known || ((padtiles >> it) & 1ull) != 0);
    const bool tile_full = QS || (!RP && !DROP && !tile_pad && (k0 + 64 <= Tk) && (!CAUSAL || k0 + 63 <= qw0) && lo_wave_max <= k0 &&
                                  k0 + 64 <= hi_wave_min);
    float alpha[QF];
    bool moved = false;
    // interior tiles first try the running maximum as it is: P = exp2(s * sl2 - m) with no tile maximum at all (32 v_max and
    // two cross-group reductions per tile less).  A lane's partial row sum above 64 (so no single P above 64; +inf on the first
    // tile, where m = -inf) sends the whole wave through the ordinary path below, which moves the maximum.
    // (bidirectional kernels only: a causal wave of the LLM shape sees three or four tiles, the first of them always slow)
    bool fast_done = false;
    // P of this tile.  Every path below writes it here and the second product packs it from here: the raw scores `s` stay intact for
    // the ordinary path when the attempt fails, and a successful attempt costs no register copies (round 5: the former `s = pt` merge
    // was 32 v_mov per tile in the Whisper loop, a quarter of its VALU instructions)
    // (kernels without the attempt, and the QS form, which computes the scores again when it fails, keep P in the registers of s)
    constexpr bool PSEP = !CAUSAL && !QS;
    f32x4_t pt_own[PSEP ? QF : 1][4];
    auto& pt = pick_ref<PSEP>(pt_own, s);
    if (!CAUSAL && tile_full && (!QS || pre)) {
      float rs[QF];
      bool over = false;
#pragma unroll
      for (int f = 0; f < QF; f++) {
        float acc0 = 0.f, acc1 = 0.f;   // two chains (the translation unit is built without SLP packing: v_pk_add_f32 beside MFMAs is slower than two v_add_f32)
        if constexpr (QS) {             // P and the two chains were produced between the MFMAs of the first product (FUSE)
          acc0 = fz[f];
        } else {
#pragma unroll
          for (int kf = 0; kf < 4; kf++)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              const float pv0 = fast_exp2(fmaf(s[f][kf][r], sl2, -mrow[f]));
              const float pv1 = fast_exp2(fmaf(s[f][kf][r + 1], sl2, -mrow[f]));
              pt[f][kf][r] = pv0;
              pt[f][kf][r + 1] = pv1;
              acc0 += pv0;
              acc1 += pv1;
            }
        }
        rs[f] = acc0 + acc1;
        over |= !(rs[f] <= 64.0f);
      }
      if (!__any(over)) {
        fast_done = true;
#pragma unroll
        for (int f = 0; f < QF; f++) {
          alpha[f] = 1.0f;
          lrow[f] += rs[f];
        }
      } else if constexpr (QS) {
        // rare (a row's maximum grew by more than 2^6 inside one tile): the ordinary path below wants the scores themselves.  They are
        // computed again from the K tile (still in its stage; no LDS read is outstanding here) instead of being kept alive beside P
        // through every successful attempt: that cost 32 VGPRs of the 168 three waves per SIMD allow
        // (its LDS addresses are derived from an opaque copy of the thread id: one register less carried through the loop for this path)
        int t2 = threadIdx.x;
        asm volatile("" : "+v"(t2));
        if (tail) s_product(false, std::true_type{}, t2 & 15, (t2 >> 4) & 3, std::false_type{});
        else s_product(false, std::false_type{}, t2 & 15, (t2 >> 4) & 3, std::false_type{});
      }
    }
    if (fast_done) {
    } else if (tile_full) {
#pragma unroll
      for (int f = 0; f < QF; f++) {
        float mt = s[f][0][0];
#pragma unroll
        for (int kf = 0; kf < 4; kf++)
#pragma unroll
          for (int r = 0; r < 4; r++) mt = fmaxf(mt, s[f][kf][r]);
        mt = max_across_groups(mt);
        mt *= ksc;
        const bool mv = mt > mrow[f] + 8.0f;
        const float mnew = mv ? mt : mrow[f];
        alpha[f] = mv ? fast_exp2(mrow[f] - mnew) : 1.0f;
        moved |= mv;
        mrow[f] = mnew;
        float rs = 0.f;
#pragma unroll
        for (int kf = 0; kf < 4; kf++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float pv = fast_exp2(fmaf(s[f][kf][r], ksc, -mnew));
            pt[f][kf][r] = pv;
            rs += pv;
          }
        lrow[f] = lrow[f] * alpha[f] + rs;
      }
    } else {
      // boundary tile: query q sees the key RANGE [qlo, min(q, qhi - 1, Tk - 1)] -- one unsigned compare per element -- minus
      // the padded keys, whose mask bytes are only fetched for tiles that hold one (PAD)
      auto masked = [&](auto pad_c) {
        constexpr bool PAD = decltype(pad_c)::value;
        unsigned mk[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
        if constexpr (PAD) {
#pragma unroll
          for (int kf = 0; kf < 4; kf++)
            mk[kf] = *reinterpret_cast<const unsigned*>(kmask_ + (int64_t)b * Tkp + k0 + (kf >> 1) * 32 + 8 * g + (kf & 1) * 4);
        }
#pragma unroll
        for (int f = 0; f < QF; f++) {
          const int q = qw0 + f * 16 + li;
          const int hi = min(min(qhi[f], Tk) - 1, CAUSAL ? q : 0x7fffffff);
          const unsigned span = (unsigned)(hi - qlo[f]);
          const bool any = hi >= qlo[f];
          const int rel = k0 + 8 * g - qlo[f];
          float mt = -INFINITY;
#pragma unroll
          for (int kf = 0; kf < 4; kf++) {
            const int ko = (kf >> 1) * 32 + (kf & 1) * 4;   // key offset of the fragment inside the lane's run
            float bias[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (RP) {   // gate[q] * table[key - q + T - 1], in log2 units like the scores
              const float* tp = p.rp_tab + (int64_t)h * p.rp_ld + (k0 + 8 * g + ko - min(q, Tq - 1) + p.rp_T - 1);
#pragma unroll
              for (int r = 0; r < 4; r++) bias[r] = rp_g[f] * tp[r];
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
              bool ok = any && (unsigned)(rel + ko + r) <= span;
              if constexpr (PAD) ok = ok && (mk[kf] & (0xffu << (8 * r))) != 0;
              const float x = ok ? fmaf(s[f][kf][r], ksc, bias[r]) : -INFINITY;
              s[f][kf][r] = x;
              mt = fmaxf(mt, x);
            }
          }
          mt = max_across_groups(mt);
          const bool mv = mt > mrow[f] + 8.0f;            // (-inf > -inf + 8 is false: a row that has seen no key stays at -inf)
          const float mnew = mv ? mt : mrow[f];
          const float muse = (mnew == -INFINITY) ? 0.f : mnew;
          alpha[f] = mv ? fast_exp2(mrow[f] - muse) : 1.0f;
          moved |= mv;
          mrow[f] = mnew;
          float rs = 0.f;
#pragma unroll
          for (int kf = 0; kf < 4; kf++) {
            unsigned keep = 0xFu;
            if constexpr (DROP) keep = attn_keep4(p, b * p.Hq + h, min(q, Tq - 1), k0 + (kf >> 1) * 32 + 8 * g + (kf & 1) * 4);
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const float pv = fast_exp2(s[f][kf][r] - muse);
              rs += pv;                                    // the row sum is over the UNdropped probabilities
              pt[f][kf][r] = DROP ? (((keep >> r) & 1u) ? pv * p.drop_scale : 0.f) : pv;
            }
          }
          lrow[f] = lrow[f] * alpha[f] + rs;
        }
      };
      if (tile_pad) masked(std::true_type{});
      else masked(std::false_type{});
    }
    if (__any(moved)) {
#pragma unroll
      for (int f = 0; f < QF; f++)
#pragma unroll
        for (int df = 0; df < DF; df++)
#pragma unroll
          for (int r = 0; r < 4; r++) o[f][df][r] *= alpha[f];
    }
    stamp(it, 5);
    // ---- O^T += V^T . P^T ----
    if constexpr (TRV) {
      frag_t pb[2][QF];
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int f = 0; f < QF; f++) pb[a][f] = pack_frag(pt[f][2 * a], pt[f][2 * a + 1]);
      static_for<TPRE, TAH>([&](auto t) {      // (PLAIN: nothing was requested before the softmax)
        vtr[t].lo = lds_read_tr<((t / DF) * 32) * KROWB>(vta[t % DF]);
        vtr[t].hi = lds_read_tr<((t / DF) * 32 + 4) * KROWB>(vta[t % DF]);
      });
      static_for<0, NT>([&](auto t) {
        constexpr int inflight = (NT - t < TAH ? NT - t : TAH);   // fragments requested and not yet waited for, incl. t
        lds_wait<2 * (inflight - 1)>(vtr[t]);
        if constexpr (t + TAH < NT) {
          vtr[t + TAH].lo = lds_read_tr<(((t + TAH) / DF) * 32) * KROWB>(vta[(t + TAH) % DF]);
          vtr[t + TAH].hi = lds_read_tr<(((t + TAH) / DF) * 32 + 4) * KROWB>(vta[(t + TAH) % DF]);
        }
        const frag_t av = tr_join(vtr[t]);
#pragma unroll
        for (int f = 0; f < QF; f++) o[f][t % DF] = mfma16(av, pb[t / DF][f], o[f][t % DF]);
      });
    } else {
    if constexpr (PLAIN) static_for<0, DF>([&](auto df) { vfr[0][df] = lds_read128<df * 16 * 128>(va[0]); });
    static_for<0, 2>([&](auto a) {
      frag_t pb[QF];
#pragma unroll
      for (int f = 0; f < QF; f++) pb[f] = pack_frag(pt[f][2 * a], pt[f][2 * a + 1]);
      if constexpr (a == 0) static_for<0, DF>([&](auto df) { vfr[1][df] = lds_read128<df * 16 * 128>(va[1]); });
      static_for<0, DF>([&](auto df) {
        lds_wait<(a == 0 ? DF : 0) + (DF - 1 - df)>(vfr[a][df]);
#pragma unroll
        for (int f = 0; f < QF; f++) o[f][df] = mfma16(vfr[a][df], pb[f], o[f][df]);
      });
    });
    }
    stamp(it, 6);
  }
  if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail DMAs still target this workgroup's LDS

  // ---- epilogue ----
  // (QS: the lane's row / column indices are derived again from an opaque copy of the thread id, so that the output addresses are
  // computed here instead of being carried -- spilled, at the 168 registers of three waves per SIMD -- from the prologue through the loop)
  int e_li = li, e_g = g, e_qw0 = qw0;
  if constexpr (QS) {
    int t2 = threadIdx.x;
    asm volatile("" : "+v"(t2));
    e_li = t2 & 15;
    e_g = (t2 >> 4) & 3;
    e_qw0 = qb0 + __builtin_amdgcn_readfirstlane(t2 >> 6) * QW;
  }
#pragma unroll
  for (int f = 0; f < QF; f++) {
    float lt = lrow[f];
    lt = sum_across_groups(lt);
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    const int q = e_qw0 + f * 16 + e_li;
    if (q >= Tq) continue;
    bf16_t* orow = p.O + ((int64_t)b * Tq + q) * p.ldo + h * D;
#pragma unroll
    for (int df = 0; df < DF; df++) {
      uint2 w;
      w.x = pack2bf(o[f][df][0] * inv, o[f][df][1] * inv);
      w.y = pack2bf(o[f][df][2] * inv, o[f][df][3] * inv);
      *reinterpret_cast<uint2*>(orow + df * 16 + 4 * e_g) = w;
    }
    if (p.LSE && e_g == 0)
      p.LSE[((int64_t)b * p.Hq + h) * Tqp + q] = lt > 0.f ? (mrow[f] * LN2 + __logf(lt)) : INFINITY;
  }
}

// ------------------------------------------------------------------------------------------
// backward dQ: workgroup = 4 waves x QF x 16 queries; 32-key K / V / K^T tiles are staged through LDS (shared by the
// four waves, next tile prefetched into registers during the MFMA phase)
//   S^T = K Q^T, dP^T = V dO^T, dS^T = P^T o (dP^T - Delta) * scale, dQ^T += K^T(as [d x keys]) . dS^T
// Row i of key fragment kf is key 8 (i / 4) + 4 kf + i % 4 of the tile, so the eight dS^T values a lane packs for the
// last product are keys 8 g .. 8 g + 7 and its K^T operand is ONE 16-byte LDS read.  With QF = 2 every K / V / K^T
// fragment read from LDS feeds two MFMAs (at QF = 1 the kernel issues one b128 read per MFMA and is LDS-bound).
// ------------------------------------------------------------------------------------------
template <int D, bool CAUSAL, bool DROP = false, int QF = 1, bool RP = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnParams p) {
  constexpr int KD = D / 32;
  constexpr int DF = D / 16;
  constexpr int ROWB = D * 2;
  constexpr int KCH = D / 8;
  constexpr int KCM = KCH - 1;
  constexpr int NKV = 32 * KCH / 256;  // 16-byte chunks per thread of a [32][D] tile
  constexpr int NKT = D * 4 / 256;     // 16-byte chunks per thread of the [D][32] transposed tile
  constexpr int QB = 64 * QF;          // queries per workgroup
  __shared__ __attribute__((aligned(16))) char lds[2 * 32 * ROWB + D * 64];
  __shared__ unsigned ldsMask[8];  // key-padding mask bytes of the current 32-key tile
  char* ldsK = lds;
  char* ldsV = lds + 32 * ROWB;
  char* ldsKt = lds + 64 * ROWB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const AttnBlk blk = attn_blk(p);
  const int b = blk.z, h = blk.y;
  const int hk = h / (p.Hq / p.Hkv);
  const int Tq = p.Tq, Tk = p.Tk, Tqp = p.Tqp, Tkp = p.Tkp;
  const int qb0 = blk.x * QB, qw0 = qb0 + wave * 16 * QF;

  frag_t qf[QF][KD], dof[QF][KD];
  float lse2[QF], delta[QF];
  int qlo[QF];
#pragma unroll
  for (int f = 0; f < QF; f++) {
    const int q = qw0 + f * 16 + li;
    const bool qok = q < Tq;
#pragma unroll
    for (int kd = 0; kd < KD; kd++) {
      qf[f][kd] = qok ? *reinterpret_cast<const frag_t*>(p.Q + ((int64_t)b * Tq + q) * p.ldq + h * D + kd * 32 + g * 8) : zero_frag();
      dof[f][kd] = qok ? *reinterpret_cast<const frag_t*>(p.dO + ((int64_t)b * Tq + q) * p.lddo + h * D + kd * 32 + g * 8) : zero_frag();
    }
    lse2[f] = qok ? p.LSE[((int64_t)b * p.Hq + h) * Tqp + q] * LOG2E : INFINITY;
    // Delta[q] = sum_d dO[q,d] O[q,d]: this lane already holds a quarter of dO's row; the 4 lanes of a row (li + 16 g)
    // combine theirs.  Written out for the dK/dV kernel that runs next on the stream (no separate delta pass over O, dO).
    float dl = 0.f;
    if (qok) {
#pragma unroll
      for (int kd = 0; kd < KD; kd++) {
        const frag_t of = *reinterpret_cast<const frag_t*>(p.O + ((int64_t)b * Tq + q) * p.ldo + h * D + kd * 32 + g * 8);
        const u16x8_t ov = __builtin_bit_cast(u16x8_t, of), dv = __builtin_bit_cast(u16x8_t, dof[f][kd]);
#pragma unroll
        for (int e = 0; e < 8; e++) dl = fmaf(bf2f(ov[e]), bf2f(dv[e]), dl);
      }
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    if (qok && g == 0) p.Delta[((int64_t)b * p.Hq + h) * Tqp + q] = dl;
    delta[f] = dl;
    qlo[f] = p.seg_lo ? p.seg_lo[(int64_t)b * Tq + min(q, Tq - 1)] : 0;
  }
  const float sl2 = p.scale * LOG2E;
  float rp_g[QF];   // gate[b, h, q] in log2 units (RP: gated relative position bias, as in the forward)
#pragma unroll
  for (int f = 0; f < QF; f++)
    rp_g[f] = RP ? p.rp_gate[((int64_t)b * p.Hq + h) * Tqp + min(qw0 + f * 16 + li, Tq - 1)] * LOG2E : 0.f;

  f32x4_t dq[QF][DF];
#pragma unroll
  for (int f = 0; f < QF; f++)
#pragma unroll
    for (int df = 0; df < DF; df++) dq[f][df] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int kend = CAUSAL ? min(Tk, qb0 + QB) : Tk;
  const int ntiles = (kend + 31) / 32;
  const bf16_t* ktb = p.Kt + ((int64_t)(b * p.Hkv + hk) * D) * Tkp;
  const int tbeg = p.seg_lo ? p.seg_lo[(int64_t)b * Tq + min(qb0, Tq - 1)] / 32 : 0;   // packed batches (see AttnParams)

  frag_t kreg[NKV], vreg[NKV], ktreg[NKT];
  unsigned mreg = 0x01010101u;
  auto gload = [&](int k0) {
    if (p.kmask && tid < 8) mreg = *reinterpret_cast<const unsigned*>(p.kmask + (int64_t)b * Tkp + k0 + tid * 4);
#pragma unroll
    for (int i = 0; i < NKV; i++) {
      const int item = tid + i * 256;
      const int row = item / KCH, c = item % KCH;
      const int key = k0 + row;
      const bool ok = key < Tk;
      kreg[i] = ok ? *reinterpret_cast<const frag_t*>(p.K + ((int64_t)b * Tk + key) * p.ldk + hk * D + c * 8) : zero_frag();
      vreg[i] = ok ? *reinterpret_cast<const frag_t*>(p.V + ((int64_t)b * Tk + key) * p.ldv + hk * D + c * 8) : zero_frag();
    }
#pragma unroll
    for (int i = 0; i < NKT; i++) {
      const int item = tid + i * 256;
      const int d = item >> 2, c = item & 3;
      ktreg[i] = *reinterpret_cast<const frag_t*>(ktb + (int64_t)d * Tkp + k0 + c * 8);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NKV; i++) {
      const int item = tid + i * 256;
      const int row = item / KCH, c = item % KCH;
      const int off = row * ROWB + ((c ^ (row & KCM)) << 4);
      *reinterpret_cast<frag_t*>(ldsK + off) = kreg[i];
      *reinterpret_cast<frag_t*>(ldsV + off) = vreg[i];
    }
#pragma unroll
    for (int i = 0; i < NKT; i++) {
      const int item = tid + i * 256;
      const int d = item >> 2, c = item & 3;
      *reinterpret_cast<frag_t*>(ldsKt + d * 64 + ((c ^ ((d >> 2) & 3)) << 4)) = ktreg[i];
    }
    if (tid < 8) ldsMask[tid] = mreg;
  };

  // RP (WavLM's gated bias, un-frozen encoder): d(gate)[q] = sum_k dS[q,k] table[k - q] is a CANCELLING sum (sum_k dS = 0 exactly), so
  // a Delta that is not the one the recomputed P implies -- sum_d dO O carries the bf16 rounding of O -- leaks Delta_err * sum_k P table
  // straight into it.  The RP form therefore walks the key tiles twice: pass 0 only accumulates sum_k P dP and sum_k P from the SAME
  // recomputed values pass 1 uses (same instructions, same bits), Delta := their quotient replaces the prologue's value (for this
  // kernel, and through p.Delta for the dK / dV kernel), and sum_k dS of pass 1 is zero to fp32 rounding.  Non-RP forms: one pass.
  float dsum[QF], psum[QF];
#pragma unroll
  for (int f = 0; f < QF; f++) dsum[f] = psum[f] = 0.f;
  for (int pass = RP ? 0 : 1; pass < 2; pass++) {
  if (ntiles > tbeg) gload(tbeg * 32);
  for (int it = tbeg; it < ntiles; it++) {
    const int k0 = it * 32;
    __syncthreads();
    lstore();
    __syncthreads();
    if (it + 1 < ntiles) gload(k0 + 32);
    if (qw0 >= Tq || (CAUSAL && k0 > qw0 + 16 * QF - 1)) continue;

    f32x4_t st[QF][2], dpt[QF][2];
#pragma unroll
    for (int kf = 0; kf < 2; kf++) {
      const int row = 8 * (li >> 2) + 4 * kf + (li & 3);
#pragma unroll
      for (int f = 0; f < QF; f++) {
        st[f][kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        dpt[f][kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int kd = 0; kd < KD; kd++) {
        const int off = row * ROWB + (((kd * 4 + g) ^ (row & KCM)) << 4);
        const frag_t kfr = *reinterpret_cast<const frag_t*>(ldsK + off);
        const frag_t vfr = *reinterpret_cast<const frag_t*>(ldsV + off);
#pragma unroll
        for (int f = 0; f < QF; f++) {
          st[f][kf] = mfma16(kfr, qf[f][kd], st[f][kf]);
          dpt[f][kf] = mfma16(vfr, dof[f][kd], dpt[f][kf]);
        }
      }
    }
    frag_t dsb[QF];
    const unsigned mk2 = ldsMask[2 * g], mk3 = ldsMask[2 * g + 1];
#pragma unroll
    for (int f = 0; f < QF; f++) {
      const int q = qw0 + f * 16 + li;
      const bool qok = q < Tq;
#pragma unroll
      for (int kf = 0; kf < 2; kf++) {
        const int kb = k0 + 8 * g + 4 * kf;
        const unsigned mk = kf ? mk3 : mk2;
        unsigned keep = 0xFu;
        if constexpr (DROP) keep = attn_keep4(p, b * p.Hq + h, min(q, Tq - 1), kb);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int key = kb + r;
          const bool ok = ((mk >> (8 * r)) & 0xffu) != 0 && key < Tk && (!CAUSAL || key <= q) && qok && key >= qlo[f];
          float bias = 0.f;
          if constexpr (RP) bias = rp_g[f] * p.rp_tab[(int64_t)h * p.rp_ld + (key - min(q, Tq - 1) + p.rp_T - 1)];
          const float pv = ok ? fast_exp2(fmaf(st[f][kf][r], sl2, bias) - lse2[f]) : 0.f;
          const float dpm = DROP ? (((keep >> r) & 1u) ? dpt[f][kf][r] * p.drop_scale : 0.f) : dpt[f][kf][r];   // d(dropped P) -> dP
          if constexpr (RP) {
            if (pass == 0) {
              dsum[f] = fmaf(pv, dpm, dsum[f]);
              psum[f] += pv;
              continue;
            }
          }
          const float ds0 = pv * (dpm - delta[f]);      // dL/d(score) of (q, key)
          if constexpr (RP) {
            if (qok && key < Tkp) p.rp_ds[(((int64_t)b * p.Hq + h) * Tq + q) * Tkp + key] = ds0;
          }
          st[f][kf][r] = ds0 * p.scale;
        }
      }
      dsb[f] = pack_frag(st[f][0], st[f][1]);
    }
    if constexpr (RP) {
      if (pass == 0) continue;
    }
#pragma unroll
    for (int df = 0; df < DF; df++) {
      const int d = df * 16 + li;
      const frag_t ktf = *reinterpret_cast<const frag_t*>(ldsKt + d * 64 + ((g ^ ((d >> 2) & 3)) << 4));
#pragma unroll
      for (int f = 0; f < QF; f++) dq[f][df] = mfma16(ktf, dsb[f], dq[f][df]);
    }
  }
  if constexpr (RP) {
    if (pass == 0) {     // the four lane groups of a query row hold disjoint key subsets: fixed-order combine, then Delta := sum P dP / sum P
      __syncthreads();   // (every wave is done reading the last tile before pass 1 re-stages tile tbeg)
#pragma unroll
      for (int f = 0; f < QF; f++) {
        float a = dsum[f], s1 = psum[f];
        a += __shfl_xor(a, 16, 64);  s1 += __shfl_xor(s1, 16, 64);
        a += __shfl_xor(a, 32, 64);  s1 += __shfl_xor(s1, 32, 64);
        delta[f] = s1 > 0.f ? a / s1 : 0.f;
        const int q = qw0 + f * 16 + li;
        if (q < Tq && g == 0) p.Delta[((int64_t)b * p.Hq + h) * Tqp + q] = delta[f];
      }
    }
  }
  }
#pragma unroll
  for (int f = 0; f < QF; f++) {
    const int q = qw0 + f * 16 + li;
    if (q >= Tq) continue;
    if (p.rope_cos) rope_grad_inplace<DF>(dq[f], p.rope_cos, p.rope_sin, p.rope_pos ? p.rope_pos[(int64_t)b * Tq + q] : q, D, g);
    bf16_t* orow = p.dQ + ((int64_t)b * Tq + q) * p.lddq + h * D;
#pragma unroll
    for (int df = 0; df < DF; df++) {
      uint2 w;
      w.x = pack2bf(dq[f][df][0], dq[f][df][1]);
      w.y = pack2bf(dq[f][df][2], dq[f][df][3]);
      *reinterpret_cast<uint2*>(orow + df * 16 + 4 * g) = w;
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward dK/dV: workgroup = 4 waves x 16 keys; for every query head of the GQA group and every 32-query tile the
// Q / dO tiles (row-major, for S and dP) and Q^T / dO^T tiles (for the reductions over queries) are staged through
// LDS and shared by the four waves;  S = Q K^T, dP = dO V^T (lane owns key (l&15), 4 consecutive queries),
//   dV^T += dO^T(as [d x q]) . P,   dK^T += Q^T(as [d x q]) . dS
// ------------------------------------------------------------------------------------------
template <int D, bool CAUSAL, bool DROP = false, bool RP = false>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(AttnParams p) {
  constexpr int KD = D / 32;
  constexpr int DF = D / 16;
  constexpr int ROWB = D * 2;
  constexpr int KCH = D / 8;
  constexpr int KCM = KCH - 1;
  constexpr int NQ = 32 * KCH / 256;
  constexpr int NQT = D * 4 / 256;
  __shared__ __attribute__((aligned(16))) char lds[2 * 32 * ROWB + 2 * D * 64];
  __shared__ __attribute__((aligned(16))) float ldsLD[64];  // [0,32): LSE of the tile's queries, [32,64): Delta
  char* ldsQ = lds;
  char* ldsDO = lds + 32 * ROWB;
  char* ldsQt = lds + 64 * ROWB;
  char* ldsDOt = lds + 64 * ROWB + D * 64;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const AttnBlk blk = attn_blk(p);
  const int b = blk.z, hk = blk.y;
  const int G = p.Hq / p.Hkv;
  const int Tq = p.Tq, Tk = p.Tk, Tqp = p.Tqp, Tkp = p.Tkp;
  const int kb0 = blk.x * 64, kw0 = kb0 + wave * 16;
  const int key = kw0 + li;
  const bool kok = key < Tk && (!p.kmask || p.kmask[(int64_t)b * Tkp + min(key, Tkp - 1)] != 0);

  frag_t kf[KD], vf[KD];
#pragma unroll
  for (int kd = 0; kd < KD; kd++) {
    const bool inb = key < Tk;
    kf[kd] = inb ? *reinterpret_cast<const frag_t*>(p.K + ((int64_t)b * Tk + key) * p.ldk + hk * D + kd * 32 + g * 8) : zero_frag();
    vf[kd] = inb ? *reinterpret_cast<const frag_t*>(p.V + ((int64_t)b * Tk + key) * p.ldv + hk * D + kd * 32 + g * 8) : zero_frag();
  }
  f32x4_t dk[DF], dv[DF];
#pragma unroll
  for (int df = 0; df < DF; df++) {
    dk[df] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    dv[df] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const float sl2 = p.scale * LOG2E;
  const int qstart = CAUSAL ? (kb0 / 32) * 32 : 0;
  // packed batches: queries at or beyond the end of the tile's last sequence never see these keys
  const int qend = p.seg_hi ? min(Tq, p.seg_hi[(int64_t)b * Tk + min(kb0 + 63, Tk - 1)]) : Tq;
  const int khi = p.seg_hi ? p.seg_hi[(int64_t)b * Tk + min(key, Tk - 1)] : 0x7fffffff;
  const int nq = max(0, (qend - qstart + 31) / 32);
  const int ntiles = G * nq;

  frag_t qreg[NQ], doreg[NQ], qtreg[NQT], dotreg[NQT];
  float ldreg = 0.f;
  auto gload = [&](int it) {
    const int hh = it / nq, q0 = qstart + (it - hh * nq) * 32;
    const int h = hk * G + hh;
    // per-query softmax statistics ride along with the tile (a global load inside the MFMA loop sat on the critical
    // path of every iteration: 2 waves per SIMD cannot hide it)
    if (tid < 64) {
      const float* src = (tid < 32 ? p.LSE : p.Delta) + ((int64_t)b * p.Hq + h) * Tqp;
      const int qq = q0 + (tid & 31);
      ldreg = qq < Tqp ? src[qq] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NQ; i++) {
      const int item = tid + i * 256;
      const int row = item / KCH, c = item % KCH;
      const int q = q0 + row;
      const bool ok = q < Tq;
      qreg[i] = ok ? *reinterpret_cast<const frag_t*>(p.Q + ((int64_t)b * Tq + q) * p.ldq + h * D + c * 8) : zero_frag();
      doreg[i] = ok ? *reinterpret_cast<const frag_t*>(p.dO + ((int64_t)b * Tq + q) * p.lddo + h * D + c * 8) : zero_frag();
    }
    const int64_t tb = ((int64_t)(b * p.Hq + h) * D) * Tqp + q0;
#pragma unroll
    for (int i = 0; i < NQT; i++) {
      const int item = tid + i * 256;
      const int d = item >> 2, c = item & 3;
      qtreg[i] = *reinterpret_cast<const frag_t*>(p.Qt + tb + (int64_t)d * Tqp + c * 8);
      dotreg[i] = *reinterpret_cast<const frag_t*>(p.dOt + tb + (int64_t)d * Tqp + c * 8);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NQ; i++) {
      const int item = tid + i * 256;
      const int row = item / KCH, c = item % KCH;
      const int off = row * ROWB + ((c ^ (row & KCM)) << 4);
      *reinterpret_cast<frag_t*>(ldsQ + off) = qreg[i];
      *reinterpret_cast<frag_t*>(ldsDO + off) = doreg[i];
    }
#pragma unroll
    for (int i = 0; i < NQT; i++) {
      const int item = tid + i * 256;
      const int d = item >> 2, c = item & 3;
      const int off = d * 64 + ((c ^ ((d >> 2) & 3)) << 4);
      *reinterpret_cast<frag_t*>(ldsQt + off) = qtreg[i];
      *reinterpret_cast<frag_t*>(ldsDOt + off) = dotreg[i];
    }
    if (tid < 64) ldsLD[tid] = ldreg;
  };

  if (ntiles > 0) gload(0);
  for (int it = 0; it < ntiles; it++) {
    const int hh = it / nq, q0 = qstart + (it - hh * nq) * 32;
    const int h = hk * G + hh;
    __syncthreads();
    lstore();
    __syncthreads();
    if (it + 1 < ntiles) gload(it + 1);
    if (kw0 >= Tk || (CAUSAL && q0 + 31 < kw0)) continue;

    f32x4_t s[2], dp[2];
#pragma unroll
    for (int f = 0; f < 2; f++) {
      const int row = f * 16 + li;
      f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f}, c = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kd = 0; kd < KD; kd++) {
        const int off = row * ROWB + (((kd * 4 + g) ^ (row & KCM)) << 4);
        const frag_t qfr = *reinterpret_cast<const frag_t*>(ldsQ + off);
        const frag_t dfr = *reinterpret_cast<const frag_t*>(ldsDO + off);
        a = mfma16(qfr, kf[kd], a);
        c = mfma16(dfr, vf[kd], c);
      }
      s[f] = a;
      dp[f] = c;
    }
    f32x4_t pm[2], ds[2];
#pragma unroll
    for (int f = 0; f < 2; f++) {
      const int qb = q0 + f * 16 + 4 * g;
      const float4 l4 = *reinterpret_cast<const float4*>(&ldsLD[f * 16 + 4 * g]);
      const float4 d4 = *reinterpret_cast<const float4*>(&ldsLD[32 + f * 16 + 4 * g]);
      const float ls[4] = {l4.x, l4.y, l4.z, l4.w};
      const float dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int q = qb + r;
        const bool ok = kok && q < Tq && (!CAUSAL || key <= q) && q < khi;
        float bias = 0.f;
        if constexpr (RP) {
          const int qq = min(q, Tq - 1);
          bias = p.rp_gate[((int64_t)b * p.Hq + h) * Tqp + qq] * LOG2E * p.rp_tab[(int64_t)h * p.rp_ld + (min(key, Tk - 1) - qq + p.rp_T - 1)];
        }
        const float pv = ok ? fast_exp2(fmaf(s[f][r], sl2, bias) - ls[r] * LOG2E) : 0.f;
        bool kept = true;   // this lane walks QUERIES: one mask word per element (its key is bit key & 3 of the word)
        if constexpr (DROP) kept = (attn_keep4(p, b * p.Hq + h, min(q, Tq - 1), key & ~3) >> (key & 3)) & 1u;
        pm[f][r] = DROP ? (kept ? pv * p.drop_scale : 0.f) : pv;                                  // dV sees the dropped P
        const float dpm = DROP ? (kept ? dp[f][r] * p.drop_scale : 0.f) : dp[f][r];
        ds[f][r] = ok ? pv * (dpm - dl[r]) * p.scale : 0.f;
      }
    }
    const frag_t pb = pack_frag(pm[0], pm[1]);
    const frag_t dsb = pack_frag(ds[0], ds[1]);
#pragma unroll
    for (int df = 0; df < DF; df++) {
      const int d = df * 16 + li;
      const int sw = (d >> 2) & 3;
      const int olo = d * 64 + (((g >> 1) ^ sw) << 4) + (g & 1) * 8;
      const int ohi = d * 64 + (((2 + (g >> 1)) ^ sw) << 4) + (g & 1) * 8;
      const u16x4_t dlo = *reinterpret_cast<const u16x4_t*>(ldsDOt + olo);
      const u16x4_t dhi = *reinterpret_cast<const u16x4_t*>(ldsDOt + ohi);
      dv[df] = mfma16(join_frag(dlo, dhi), pb, dv[df]);
      const u16x4_t qlo = *reinterpret_cast<const u16x4_t*>(ldsQt + olo);
      const u16x4_t qhi = *reinterpret_cast<const u16x4_t*>(ldsQt + ohi);
      dk[df] = mfma16(join_frag(qlo, qhi), dsb, dk[df]);
    }
  }
  if (key >= Tk) return;
  if (p.rope_cos) rope_grad_inplace<DF>(dk, p.rope_cos, p.rope_sin, p.rope_pos ? p.rope_pos[(int64_t)b * Tk + key] : key, D, g);
  bf16_t* krow = p.dK + ((int64_t)b * Tk + key) * p.lddk + hk * D;
  bf16_t* vrow = p.dV + ((int64_t)b * Tk + key) * p.lddv + hk * D;
#pragma unroll
  for (int df = 0; df < DF; df++) {
    uint2 w;
    w.x = pack2bf(dk[df][0], dk[df][1]);
    w.y = pack2bf(dk[df][2], dk[df][3]);
    *reinterpret_cast<uint2*>(krow + df * 16 + 4 * g) = w;
    w.x = pack2bf(dv[df][0], dv[df][1]);
    w.y = pack2bf(dv[df][2], dv[df][3]);
    *reinterpret_cast<uint2*>(vrow + df * 16 + 4 * g) = w;
  }
}

// ------------------------------------------------------------------------------------------
// backward dK/dV, second form (shipped): the round-1 kernel above is LATENCY bound -- ~9.7 k cycles per 32-query tile against
// 0.5 k cycles of MFMA work per wave: one 32 KiB tile per workgroup in flight (64 KiB per CU) cannot cover the ~2 us an HBM/L2
// round trip takes under load.  Here
//   * a workgroup is 8 waves x 16 keys = 128 keys, so every staged Q / dO tile is used by twice as many keys (half the bytes
//     per FLOP);
//   * the tiles are DMA'd HBM -> LDS (global_load_lds_dwordx4, no staging registers) into a ring of 4 stages; three tiles
//     (96 KiB per CU) are in flight while the fourth is consumed; the waits are counted (vmcnt), one raw barrier per tile;
//   * the LDS images keep the XOR swizzles of the first form: the DMA destination is lane-linear, so the swizzle is applied
//     to the per-lane SOURCE chunk (same involution as on the read side).
// The arithmetic per wave and tile is unchanged (same masks, same RoPE / GQA epilogue).
// ------------------------------------------------------------------------------------------
// XOR key of the 16-byte chunks of row `row` of a row-major ring sub-tile
template <int D>
__device__ __forceinline__ int ring_swz(int row) {
  const int i = ((row >> 3) << 2) | (row & 3);   // position of the row inside its fragment (0..15)
  return D == 128 ? i : (i >> 1);
}

// ABL != 0: timing ablations for tools/attn_bwd_bench.py (WRONG results): 1 no DMA inside the loop, 2 no softmax arithmetic,
// 5 no barrier
template <int D, bool CAUSAL, int ABL = 0>
__global__ __launch_bounds__(512) void attn_bwd_dkdv_ring_kernel(AttnParams p) {
  constexpr int KD = D / 32;
  constexpr int DF = D / 16;
  constexpr int ROWB = D * 2;
  constexpr int KCH = D / 8;
  constexpr int KCM = KCH - 1;
  constexpr int NS = 4;                       // ring stages
  constexpr int SUB = 32 * ROWB;              // bytes of one sub-tile: [32][D] row-major == [D][32] transposed
  constexpr int STG = 4 * SUB + 1024;         // Q | dO | Qt | dOt | LSE[32] Delta[32] (+ the rest of that DMA piece)
  constexpr int NI = SUB / 1024;              // 1 KiB DMA instructions per sub-tile (8 for D = 128, 4 for D = 64)
  constexpr int NU = 4 * NI / 8;              // tile DMA instructions per wave and stage
  constexpr int PW = NU;                      // (wave 0: + 1, the LSE / Delta line)
  constexpr int RPI = 1024 / ROWB;            // rows of a row-major sub-tile per DMA instruction (4 | 8)
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const AttnBlk blk = attn_blk(p);
  const int b = blk.z, hk = blk.y;
  const int G = p.Hq / p.Hkv;
  const int Tq = p.Tq, Tk = p.Tk, Tqp = p.Tqp, Tkp = p.Tkp;
  const int kb0 = blk.x * 128, kw0 = kb0 + wave * 16;
  const int key = kw0 + li;
  const bool kok = key < Tk && (!p.kmask || p.kmask[(int64_t)b * Tkp + min(key, Tkp - 1)] != 0);

  frag_t kf[KD], vf[KD];
#pragma unroll
  for (int kd = 0; kd < KD; kd++) {
    const bool inb = key < Tk;
    kf[kd] = inb ? *reinterpret_cast<const frag_t*>(p.K + ((int64_t)b * Tk + key) * p.ldk + hk * D + kd * 32 + g * 8) : zero_frag();
    vf[kd] = inb ? *reinterpret_cast<const frag_t*>(p.V + ((int64_t)b * Tk + key) * p.ldv + hk * D + kd * 32 + g * 8) : zero_frag();
  }
  f32x4_t dk[DF], dv[DF];
#pragma unroll
  for (int df = 0; df < DF; df++) {
    dk[df] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    dv[df] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const float sl2 = p.scale * LOG2E;
  const int qstart = CAUSAL ? (kb0 / 32) * 32 : 0;
  const int qend = p.seg_hi ? min(Tq, p.seg_hi[(int64_t)b * Tk + min(kb0 + 127, Tk - 1)]) : Tq;
  int khi = p.seg_hi ? p.seg_hi[(int64_t)b * Tk + min(key, Tk - 1)] : 0x7fffffff;
  // every ordinary load above must have RETURNED before the first asm DMA is issued: hipcc would otherwise sink its
  // s_waitcnt vmcnt(0) for them to their first use inside the tile loop, where the hardware counter also holds the DMAs
#pragma unroll
  for (int kd = 0; kd < KD; kd++) asm volatile("" : "+v"(kf[kd]), "+v"(vf[kd]));
  asm volatile("" : "+v"(khi));
  // wave-uniform facts for the mask-free path: all 16 keys of this wave exist and are attendable; first query row that some
  // key of the wave must NOT see (seg_hi is non-decreasing: the wave's first key has the smallest)
  const bool wave_all_keys = __builtin_amdgcn_readfirstlane(__all(kok ? 1 : 0));
  const int qlim = min(Tq, __builtin_amdgcn_readfirstlane(khi));
  const int nq = max(0, (qend - qstart + 31) / 32);
  const int ntiles = G * nq;

  // ---- DMA issue of one tile into one stage.  This wave owns DMA instructions idx = wave + 8u (u < NU) of the 4 * NI that
  // make a stage: sub-tile sub = idx / NI (0 Q, 1 dO, 2 Q^T, 3 dO^T), piece j = idx % NI.  Per lane ONE byte offset per owned
  // instruction is computed here; per tile only a scalar is added (the first form spent more VALU on 64-bit address math and
  // the tile-index division than on the softmax). ----
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_offset_of(lds));
  const int nB = p.gz;
  const __amdgpu_buffer_rsrc_t srd_q = __builtin_amdgcn_make_buffer_rsrc((void*)p.Q, 0, (unsigned)((((int64_t)nB * Tq - 1) * p.ldq + (int64_t)p.Hq * D) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_do = __builtin_amdgcn_make_buffer_rsrc((void*)p.dO, 0, (unsigned)((((int64_t)nB * Tq - 1) * p.lddo + (int64_t)p.Hq * D) * 2), 0x00020000);
  const unsigned tbytes = (unsigned)((int64_t)nB * p.Hq * D * Tqp * 2);
  const __amdgpu_buffer_rsrc_t srd_qt = __builtin_amdgcn_make_buffer_rsrc((void*)p.Qt, 0, tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_dot = __builtin_amdgcn_make_buffer_rsrc((void*)p.dOt, 0, tbytes, 0x00020000);
  // D = 128 (NI = 8): instruction u of this wave is piece j = wave of sub-tile u.  D = 64 (NI = 4): waves 0-3 take Q and Q^T,
  // waves 4-7 take dO and dO^T, piece j = wave & 3.  Either way the first NU / 2 are row-major, the rest transposed, and the
  // descriptor of each is fixed per wave.
  const bool second = (D == 64) && wave >= 4;
  const int jpiece = (D == 128) ? wave : (wave & 3);
  __amdgpu_buffer_rsrc_t srd_u[NU];
  unsigned voff[NU], ldu[NU], dsto[NU];
#pragma unroll
  for (int u = 0; u < NU; u++) {
    const int sub = (D == 128) ? u : 2 * u + (second ? 1 : 0);   // 0 Q, 1 dO, 2 Q^T, 3 dO^T
    const bool isdo = sub & 1;
    if (u < NU / 2) {
      const int row = jpiece * RPI + lane / KCH, c = lane % KCH;
      ldu[u] = (unsigned)(isdo ? p.lddo : p.ldq);
      srd_u[u] = isdo ? srd_do : srd_q;
      voff[u] = (unsigned)(((int64_t)b * Tq + row) * ldu[u] + ((c ^ ring_swz<D>(row)) << 3)) * 2u;
    } else {
      const int d = jpiece * 16 + (lane >> 2), c = lane & 3;
      ldu[u] = 0;
      srd_u[u] = isdo ? srd_dot : srd_qt;
      voff[u] = (unsigned)((int64_t)d * Tqp + ((c ^ ((d >> 2) & 3)) << 3)) * 2u;
    }
    dsto[u] = (unsigned)(sub * SUB + jpiece * 1024);
  }
  // the LSE / Delta line: the DMA image is lane-linear (lane l -> byte 16 l of the 1 KiB piece); lanes 0-7 carry LSE, lanes
  // 8-15 Delta (bytes 128-255), the rest repeat lane 0
  const float* ld_src = (lane < 16 && lane >= 8 ? p.Delta + (lane - 8) * 4 : p.LSE + (lane < 8 ? lane * 4 : 0));
  int i_hh = 0, i_qi = 0;   // (head of the GQA group, query tile) of the next tile to issue -- advanced incrementally, no division
  auto issue = [&](int s) {
    const int h = hk * G + i_hh, q0 = qstart + i_qi * 32;
    const unsigned st = lds0 + (unsigned)(s * STG);
    const unsigned s_row = (unsigned)q0 * 2u;                                              // x ld below
    const unsigned s_t = (unsigned)((((int64_t)b * p.Hq + h) * D) * Tqp + q0) * 2u;
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(st + dsto[u]);
      const unsigned so = (u < NU / 2) ? s_row * ldu[u] + (unsigned)(h * D) * 2u : s_t;
      bufdma16_asm(srd_u[u], voff[u] + __builtin_amdgcn_readfirstlane(so), dst);
    }
    // LSE[q0 .. q0+31] (lanes 0-7, 16 bytes each) and Delta[q0 .. q0+31] (lanes 16-23 -> +1 KiB... see ld_lane_off): wave 0
    // alone fetches them, one instruction with per-lane 64-bit addresses (every extra LDS-DMA instruction costs its wave
    // 100-200 cycles of issue, MI355X_MICROARCH "LDS-DMA piece issue cost")
    if (wave == 0) {
      const int64_t e = ((int64_t)b * p.Hq + h) * Tqp + q0;
      glds16_asm(ld_src + e, __builtin_amdgcn_readfirstlane(st + (unsigned)(4 * SUB)));
    }
    // advance to the next tile; past the end the last tile is re-fetched into a stage nobody reads (keeps the counts uniform)
    if (i_hh * nq + i_qi + 1 < ntiles) {
      if (++i_qi == nq) {
        i_qi = 0;
        ++i_hh;
      }
    }
  };

  // per-lane LDS read addresses (stage 0).  Row-major Q / dO sub-tiles: fragment f, A-operand row i = li is tile row
  // 8 * (i / 4) + 4 f + i % 4 (so that the lane's eight S^T values are the CONTIGUOUS queries 8 g .. 8 g + 7 and the second
  // product's Q^T / dO^T operand is one 16-byte LDS read); a row's 16-byte chunks are XOR-swizzled with swz(row) below, which
  // is distinct over the 16 rows of a fragment (conflict-free b128 reads).  Transposed sub-tiles: [D][32 queries], 64-byte
  // rows, chunk ^ ((d >> 2) & 3).
  unsigned aA0[KD];
  {
    const int row = 8 * (li >> 2) + (li & 3);   // + 4 f through the instruction offset
#pragma unroll
    for (int kd = 0; kd < KD; kd++) aA0[kd] = lds0 + (unsigned)(row * ROWB + (((kd * 4 + g) ^ ring_swz<D>(row)) << 4));
  }
  const unsigned aL0 = lds0 + (unsigned)(32 * g);
  const unsigned aB0 = lds0 + (unsigned)(li * 64 + ((g ^ ((li >> 2) & 3)) << 4));

  if (ntiles > 0) {
    issue(0);
    issue(1);
    issue(2);
  }
  int c_qi = 0;   // query tile of the tile being consumed
  constexpr bool PROBE = ABL == 9;   // cycle stamps of waves 0 / 7 of one workgroup (tools/attn_dq_probe.py)
  const bool prb = PROBE && blk.x == 0 && blk.y == 3 && blk.z == 5 && (wave == 0 || wave == 7);
  auto stamp = [&](int it, int i) {
    if constexpr (PROBE) {
      if (prb) {
        const unsigned long long t = __builtin_readcyclecounter();
        if (lane == 0 && it < 16) g_attn_probe[((wave ? 1 : 0) * 16 + it) * 8 + i] = t;
      }
    }
  };
  for (int it = 0; it < ntiles; it++) {
    stamp(it, 0);
    const int q0 = qstart + c_qi * 32;
    if (++c_qi == nq) c_qi = 0;
    const int s = it & (NS - 1);
    // tile `it` has landed when at most the two younger tiles' DMA of this wave are outstanding; after the barrier every
    // wave's share has, and everybody is done reading stage (it - 1) % NS, which the next DMA refills
    if constexpr (ABL != 1) {
      if (wave == 0) {
        if constexpr (PW == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        if constexpr (PW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
    }
    static_assert(PW == 4 || PW == 2, "counted waits above assume 4 (D = 128) or 2 (D = 64) tile DMA per wave and stage");
    if constexpr (ABL != 5) __builtin_amdgcn_s_barrier();
    stamp(it, 1);
    // (issuing these in the softmax stretch instead measured 2 % slower; spreading them over four points of the iteration,
    // rotated by wave, 15 % slower: a DMA instruction in the middle of the chain of reads and products stalls it)
    if constexpr (ABL != 1) issue((it + 3) & (NS - 1));
    stamp(it, 2);
    if (kw0 >= Tk || (CAUSAL && q0 + 31 < kw0)) continue;
    // ---- one tile, LDS reads and their waits placed by hand (hipcc's own order was read -> wait -> two MFMAs, sixteen times
    // per tile: every pair of products exposed a full LDS latency, and since the barrier releases all eight waves at once
    // nothing else was runnable; timing ablations in profiles/r02_attention_bwd.md).  The reads are asm so that the compiler
    // cannot re-serialise them; a wait names the registers it completes ("+v"), which orders their consumers behind it.
    const unsigned so = (unsigned)(s * STG);
    unsigned aA[KD];
#pragma unroll
    for (int kd = 0; kd < KD; kd++) aA[kd] = aA0[kd] + so;
    const unsigned aL = aL0 + so, aB = aB0 + so;
    frag_t q0f[KD], d0f[KD], q1f[KD], d1f[KD], lse0, lse1, del0, del1;
    frag_t bq[DF], bd[DF];
    static_for<0, KD>([&](auto kd) {
      q0f[kd] = lds_read128<0>(aA[kd]);
      d0f[kd] = lds_read128<SUB>(aA[kd]);
    });
    lse0 = lds_read128<4 * SUB>(aL);
    lse1 = lds_read128<4 * SUB + 16>(aL);
    del0 = lds_read128<4 * SUB + 128>(aL);
    del1 = lds_read128<4 * SUB + 144>(aL);
    f32x4_t sacc[2], dp[2];
    sacc[0] = sacc[1] = dp[0] = dp[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // fragment 0 products while fragment 1's operands are fetched (2 * KD + 4 reads outstanding at every wait)
    static_for<0, KD>([&](auto kd) {
      lds_wait<2 * KD + 2>(q0f[kd], d0f[kd]);
      q1f[kd] = lds_read128<4 * ROWB>(aA[kd]);
      d1f[kd] = lds_read128<SUB + 4 * ROWB>(aA[kd]);
      sacc[0] = mfma16(q0f[kd], kf[kd], sacc[0]);
      dp[0] = mfma16(d0f[kd], vf[kd], dp[0]);
    });
    // fragment 1 products while the first half of the dO^T / Q^T operands is fetched
    static_for<0, KD>([&](auto kd) {
      if constexpr (kd == 0) lds_wait<2 * KD - 2>(lse0, lse1, del0, del1, q1f[kd], d1f[kd]);
      else lds_wait<2 * KD - 2>(q1f[kd], d1f[kd]);
      bd[kd] = lds_read128<3 * SUB + kd * 1024>(aB);
      bq[kd] = lds_read128<2 * SUB + kd * 1024>(aB);
      sacc[1] = mfma16(q1f[kd], kf[kd], sacc[1]);
      dp[1] = mfma16(d1f[kd], vf[kd], dp[1]);
    });
    stamp(it, 3);
    // softmax arithmetic (covers the latency of those reads).  Element (f, r) of this lane is query q0 + 8g + 4f + r.
    f32x4_t pm[2], ds[2];
    const f32x4_t l4[2] = {__builtin_bit_cast(f32x4_t, lse0), __builtin_bit_cast(f32x4_t, lse1)};
    const f32x4_t e4[2] = {__builtin_bit_cast(f32x4_t, del0), __builtin_bit_cast(f32x4_t, del1)};
    // interior tile: every (key, query) pair of this wave is visible -- no masks (3 of 4 tiles at the Llama shape)
    const bool interior = wave_all_keys && q0 + 32 <= qlim && (!CAUSAL || kw0 + 15 <= q0);
    if (interior) {
#pragma unroll
      for (int f = 0; f < 2; f++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float pv = fast_exp2(__builtin_fmaf(sacc[f][r], sl2, -LOG2E * l4[f][r]));
          pm[f][r] = pv;
          ds[f][r] = pv * (dp[f][r] - e4[f][r]) * p.scale;
        }
      }
    } else {
#pragma unroll
      for (int f = 0; f < 2; f++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int q = q0 + 8 * g + 4 * f + r;
          if constexpr (ABL == 2) {
            pm[f][r] = sacc[f][r];
            ds[f][r] = dp[f][r];
            continue;
          }
          const bool ok = kok && q < Tq && (!CAUSAL || key <= q) && q < khi;
          const float pv = ok ? fast_exp2(__builtin_fmaf(sacc[f][r], sl2, -LOG2E * l4[f][r])) : 0.f;
          pm[f][r] = pv;
          ds[f][r] = ok ? pv * (dp[f][r] - e4[f][r]) * p.scale : 0.f;
        }
      }
    }
    const frag_t pb = pack_frag(pm[0], pm[1]);
    const frag_t dsb = pack_frag(ds[0], ds[1]);
    stamp(it, 4);
    // dV / dK products, operands KD fragments ahead
    static_for<0, DF>([&](auto df) {
      if constexpr (df + KD < DF) {
        bd[df + KD] = lds_read128<3 * SUB + (df + KD) * 1024>(aB);
        bq[df + KD] = lds_read128<2 * SUB + (df + KD) * 1024>(aB);
        lds_wait<2 * KD>(bd[df], bq[df]);
      } else {
        lds_wait<2 * (DF - df - 1)>(bd[df], bq[df]);
      }
      dv[df] = mfma16(bd[df], pb, dv[df]);
      dk[df] = mfma16(bq[df], dsb, dk[df]);
    });
    stamp(it, 5);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail DMAs still target this workgroup's LDS
  if (key >= Tk) return;
  if (p.rope_cos) rope_grad_inplace<DF>(dk, p.rope_cos, p.rope_sin, p.rope_pos ? p.rope_pos[(int64_t)b * Tk + key] : key, D, g);
  bf16_t* krow = p.dK + ((int64_t)b * Tk + key) * p.lddk + hk * D;
  bf16_t* vrow = p.dV + ((int64_t)b * Tk + key) * p.lddv + hk * D;
#pragma unroll
  for (int df = 0; df < DF; df++) {
    uint2 w;
    w.x = pack2bf(dk[df][0], dk[df][1]);
    w.y = pack2bf(dk[df][2], dk[df][3]);
    *reinterpret_cast<uint2*>(krow + df * 16 + 4 * g) = w;
    w.x = pack2bf(dv[df][0], dv[df][1]);
    w.y = pack2bf(dv[df][2], dv[df][3]);
    *reinterpret_cast<uint2*>(vrow + df * 16 + 4 * g) = w;
  }
}


// ------------------------------------------------------------------------------------------
// backward dK/dV, transposed-read form (round 4, shipped): attn_bwd_dkdv_ring_kernel without the Q^T / dO^T sub-tiles.  A stage is
// Q | dO | LSE,Delta (17 KiB at D = 128, was 33): half the LDS-DMA pieces per wave and tile, and the [B, H, D, Tp] copies of Q and dO
// are no longer read (or written).  dO^T / Q^T operands of the second products: two ds_read_b64_tr_b16 per fragment from the row-major
// sub-tiles the first products read with ds_read_b128.  Same arithmetic, masks, RoPE / GQA epilogue and ring discipline (4 stages, three
// tiles in flight, counted vmcnt, one raw barrier per tile) as the ring kernel.
// ------------------------------------------------------------------------------------------
template <int D, bool CAUSAL>
__global__ __launch_bounds__(512) void attn_bwd_dkdv_tr_kernel(AttnParams p) {
  constexpr int KD = D / 32;
  constexpr int DF = D / 16;
  constexpr int ROWB = D * 2;
  constexpr int KCH = D / 8;
  constexpr int NS = 4;                       // ring stages
  constexpr int SUB = 32 * ROWB;              // bytes of one row-major sub-tile [32][D]
  constexpr int STG = 2 * SUB + 1024;         // Q | dO | LSE[32] Delta[32] (+ the rest of that DMA piece)
  constexpr int NI = SUB / 1024;              // 1 KiB DMA instructions per sub-tile (8 for D = 128, 4 for D = 64)
  constexpr int NU = 2 * NI / 8;              // tile DMA instructions per wave and stage: 2 (D = 128: piece `wave` of Q and of dO) | 1
  constexpr int RPI = 1024 / ROWB;            // rows per DMA instruction (4 | 8)
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const AttnBlk blk = attn_blk(p);
  const int b = blk.z, hk = blk.y;
  const int G = p.Hq / p.Hkv;
  const int Tq = p.Tq, Tk = p.Tk, Tqp = p.Tqp, Tkp = p.Tkp;
  const int kb0 = blk.x * 128, kw0 = kb0 + wave * 16;
  const int key = kw0 + li;
  const bool kok = key < Tk && (!p.kmask || p.kmask[(int64_t)b * Tkp + min(key, Tkp - 1)] != 0);

  frag_t kf[KD], vf[KD];
#pragma unroll
  for (int kd = 0; kd < KD; kd++) {
    const bool inb = key < Tk;
    kf[kd] = inb ? *reinterpret_cast<const frag_t*>(p.K + ((int64_t)b * Tk + key) * p.ldk + hk * D + kd * 32 + g * 8) : zero_frag();
    vf[kd] = inb ? *reinterpret_cast<const frag_t*>(p.V + ((int64_t)b * Tk + key) * p.ldv + hk * D + kd * 32 + g * 8) : zero_frag();
  }
  f32x4_t dk[DF], dv[DF];
#pragma unroll
  for (int df = 0; df < DF; df++) {
    dk[df] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    dv[df] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const float sl2 = p.scale * LOG2E;
  const int qstart = CAUSAL ? (kb0 / 32) * 32 : 0;
  const int qend = p.seg_hi ? min(Tq, p.seg_hi[(int64_t)b * Tk + min(kb0 + 127, Tk - 1)]) : Tq;
  int khi = p.seg_hi ? p.seg_hi[(int64_t)b * Tk + min(key, Tk - 1)] : 0x7fffffff;
  // every ordinary load above must have RETURNED before the first asm DMA is issued (see attn_bwd_dkdv_ring_kernel)
#pragma unroll
  for (int kd = 0; kd < KD; kd++) asm volatile("" : "+v"(kf[kd]), "+v"(vf[kd]));
  asm volatile("" : "+v"(khi));
  const bool wave_all_keys = __builtin_amdgcn_readfirstlane(__all(kok ? 1 : 0));
  const int qlim = min(Tq, __builtin_amdgcn_readfirstlane(khi));
  const int nq = max(0, (qend - qstart + 31) / 32);
  const int ntiles = G * nq;

  // ---- DMA issue: D = 128: this wave owns piece j = wave (rows 4 wave .. + 3) of Q (u = 0) and of dO (u = 1); D = 64: waves 0-3 own
  // piece wave (rows 8 wave .. + 7) of Q, waves 4-7 piece wave - 4 of dO.  Lane -> (row, 16-byte chunk); the swizzle goes on the SOURCE chunk.
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_offset_of(lds));
  const int nB = p.gz;
  const __amdgpu_buffer_rsrc_t srd_q = __builtin_amdgcn_make_buffer_rsrc((void*)p.Q, 0, (unsigned)((((int64_t)nB * Tq - 1) * p.ldq + (int64_t)p.Hq * D) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_do = __builtin_amdgcn_make_buffer_rsrc((void*)p.dO, 0, (unsigned)((((int64_t)nB * Tq - 1) * p.lddo + (int64_t)p.Hq * D) * 2), 0x00020000);
  const bool second = (D == 64) && wave >= 4;
  const int jpiece = (D == 128) ? wave : (wave & 3);
  __amdgpu_buffer_rsrc_t srd_u[NU];
  unsigned voff[NU], ldu[NU], dsto[NU];
#pragma unroll
  for (int u = 0; u < NU; u++) {
    const bool isdo = (D == 128) ? (u == 1) : second;
    const int row = jpiece * RPI + lane / KCH, c = lane % KCH;
    ldu[u] = (unsigned)(isdo ? p.lddo : p.ldq);
    srd_u[u] = isdo ? srd_do : srd_q;
    voff[u] = (unsigned)(((int64_t)b * Tq + row) * ldu[u] + ((c ^ tr_swz<D>(row)) << 3)) * 2u;
    dsto[u] = (unsigned)((isdo ? SUB : 0) + jpiece * 1024);
  }
  const float* ld_src = (lane < 16 && lane >= 8 ? p.Delta + (lane - 8) * 4 : p.LSE + (lane < 8 ? lane * 4 : 0));
  int i_hh = 0, i_qi = 0;
  auto issue = [&](int s) {
    const int h = hk * G + i_hh, q0 = qstart + i_qi * 32;
    const unsigned st = lds0 + (unsigned)(s * STG);
    const unsigned s_row = (unsigned)q0 * 2u;
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(st + dsto[u]);
      const unsigned so = s_row * ldu[u] + (unsigned)(h * D) * 2u;
      bufdma16_asm(srd_u[u], voff[u] + __builtin_amdgcn_readfirstlane(so), dst);
    }
    if (wave == 0) {
      const int64_t e = ((int64_t)b * p.Hq + h) * Tqp + q0;
      glds16_asm(ld_src + e, __builtin_amdgcn_readfirstlane(st + (unsigned)(2 * SUB)));
    }
    if (i_hh * nq + i_qi + 1 < ntiles) {
      if (++i_qi == nq) {
        i_qi = 0;
        ++i_hh;
      }
    }
  };

  // per-lane LDS read addresses (stage 0): first products = fragment rows 8 (li / 4) + 4 f + li % 4, chunk (4 kd + g) ^ key;
  // second products = transposed reads (tr_lane_off)
  unsigned aA0[KD];
  {
    const int row = 8 * (li >> 2) + (li & 3);   // + 4 f through the instruction offset (the key ignores bit 2 of the row)
#pragma unroll
    for (int kd = 0; kd < KD; kd++) aA0[kd] = lds0 + (unsigned)(row * ROWB + (((kd * 4 + g) ^ tr_swz<D>(row)) << 4));
  }
  const unsigned aL0 = lds0 + (unsigned)(32 * g);
  const unsigned aT0 = lds0 + tr_lane_off<D>(g, li);

  if (ntiles > 0) {
    issue(0);
    issue(1);
    issue(2);
  }
  int c_qi = 0;
  constexpr int AH = 3;        // transposed fragments (4 reads each) requested ahead of their products
  for (int it = 0; it < ntiles; it++) {
    const int q0 = qstart + c_qi * 32;
    if (++c_qi == nq) c_qi = 0;
    const int s = it & (NS - 1);
    // tile `it` has landed when at most the two younger tiles' DMA of this wave are outstanding
    if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NU + 2) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NU) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue((it + 3) & (NS - 1));
    if (kw0 >= Tk || (CAUSAL && q0 + 31 < kw0)) continue;
    const unsigned so = (unsigned)(s * STG);
    unsigned aA[KD];
#pragma unroll
    for (int kd = 0; kd < KD; kd++) aA[kd] = aA0[kd] + so;
    const unsigned aL = aL0 + so;
    unsigned aT[DF];
#pragma unroll
    for (int df = 0; df < DF; df++) aT[df] = (aT0 + so) ^ (unsigned)(df << 5);   // (so is a multiple of 1 KiB: the XOR commutes with it)
    frag_t q0f[KD], d0f[KD], q1f[KD], d1f[KD], lse0, lse1, del0, del1;
    TrFrag bq[DF], bd[DF];
    static_for<0, KD>([&](auto kd) {
      q0f[kd] = lds_read128<0>(aA[kd]);
      d0f[kd] = lds_read128<SUB>(aA[kd]);
    });
    lse0 = lds_read128<2 * SUB>(aL);
    lse1 = lds_read128<2 * SUB + 16>(aL);
    del0 = lds_read128<2 * SUB + 128>(aL);
    del1 = lds_read128<2 * SUB + 144>(aL);
    f32x4_t sacc[2], dp[2];
    sacc[0] = sacc[1] = dp[0] = dp[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // fragment 0 products while fragment 1's operands are fetched (2 KD + 4 reads outstanding at every wait)
    static_for<0, KD>([&](auto kd) {
      lds_wait<2 * KD + 2>(q0f[kd], d0f[kd]);
      q1f[kd] = lds_read128<4 * ROWB>(aA[kd]);
      d1f[kd] = lds_read128<SUB + 4 * ROWB>(aA[kd]);
      sacc[0] = mfma16(q0f[kd], kf[kd], sacc[0]);
      dp[0] = mfma16(d0f[kd], vf[kd], dp[0]);
    });
    // fragment 1 products while the first AH transposed fragments of dO / Q are fetched (four 8-byte reads each; the LDS counter
    // holds 15, so at most 12 of them are requested before the softmax)
    static_for<0, KD>([&](auto kd) {
      constexpr int before = 2 * (KD - 1 - kd) + 4 * (kd < AH ? kd : AH);   // reads younger than (q1f[kd], d1f[kd]) at this point
      if constexpr (kd == 0) lds_wait<before>(lse0, lse1, del0, del1, q1f[kd], d1f[kd]);
      else lds_wait<before>(q1f[kd], d1f[kd]);
      if constexpr (kd < AH) {
        bd[kd].lo = lds_read_tr<SUB>(aT[kd]);
        bd[kd].hi = lds_read_tr<SUB + 4 * ROWB>(aT[kd]);
        bq[kd].lo = lds_read_tr<0>(aT[kd]);
        bq[kd].hi = lds_read_tr<4 * ROWB>(aT[kd]);
      }
      sacc[1] = mfma16(q1f[kd], kf[kd], sacc[1]);
      dp[1] = mfma16(d1f[kd], vf[kd], dp[1]);
    });
    static_assert(KD >= 2 && AH <= DF, "read-ahead bookkeeping below");
    // softmax arithmetic (covers the latency of those reads).  Element (f, r) of this lane is query q0 + 8g + 4f + r.
    f32x4_t pm[2], ds[2];
    const f32x4_t l4[2] = {__builtin_bit_cast(f32x4_t, lse0), __builtin_bit_cast(f32x4_t, lse1)};
    const f32x4_t e4[2] = {__builtin_bit_cast(f32x4_t, del0), __builtin_bit_cast(f32x4_t, del1)};
    const bool interior = wave_all_keys && q0 + 32 <= qlim && (!CAUSAL || kw0 + 15 <= q0);
    if (interior) {
#pragma unroll
      for (int f = 0; f < 2; f++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float pv = fast_exp2(__builtin_fmaf(sacc[f][r], sl2, -LOG2E * l4[f][r]));
          pm[f][r] = pv;
          ds[f][r] = pv * (dp[f][r] - e4[f][r]) * p.scale;
        }
      }
    } else {
#pragma unroll
      for (int f = 0; f < 2; f++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int q = q0 + 8 * g + 4 * f + r;
          const bool ok = kok && q < Tq && (!CAUSAL || key <= q) && q < khi;
          const float pv = ok ? fast_exp2(__builtin_fmaf(sacc[f][r], sl2, -LOG2E * l4[f][r])) : 0.f;
          pm[f][r] = pv;
          ds[f][r] = ok ? pv * (dp[f][r] - e4[f][r]) * p.scale : 0.f;
        }
      }
    }
    const frag_t pb = pack_frag(pm[0], pm[1]);
    const frag_t dsb = pack_frag(ds[0], ds[1]);
    // dV / dK products, operands AH fragments ahead (wait for fragment df, then request df + AH, then its two products).  The fragments are
    // requested in fragment ORDER (the LDS returns in order and the waits below count on it): those of the top-up first (KD < AH: the first
    // products prefetched only KD of them; round 6 -- until then the top-up of fragment PRE was requested BEHIND fragment AH and waited for with a
    // count that did not cover it, correct only because hipcc happened to sink its two MFMAs below the next fragment's full wait)
    constexpr int PRE = KD < AH ? KD : AH;      // fragments requested above
    static_for<PRE, AH>([&](auto df) {
      bd[df].lo = lds_read_tr<SUB>(aT[df]);
      bd[df].hi = lds_read_tr<SUB + 4 * ROWB>(aT[df]);
      bq[df].lo = lds_read_tr<0>(aT[df]);
      bq[df].hi = lds_read_tr<4 * ROWB>(aT[df]);
    });
    static_for<0, DF>([&](auto df) {
      constexpr int inflight = (DF - df < AH ? DF - df : AH);   // fragments df .. df + inflight - 1 are requested and not yet waited for
      lds_wait<4 * (inflight - 1)>(bd[df], bq[df]);
      if constexpr (df + AH < DF) {
        bd[df + AH].lo = lds_read_tr<SUB>(aT[df + AH]);
        bd[df + AH].hi = lds_read_tr<SUB + 4 * ROWB>(aT[df + AH]);
        bq[df + AH].lo = lds_read_tr<0>(aT[df + AH]);
        bq[df + AH].hi = lds_read_tr<4 * ROWB>(aT[df + AH]);
      }
      dv[df] = mfma16(tr_join(bd[df]), pb, dv[df]);
      dk[df] = mfma16(tr_join(bq[df]), dsb, dk[df]);
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail DMAs still target this workgroup's LDS
  if (key >= Tk) return;
  if (p.rope_cos) rope_grad_inplace<DF>(dk, p.rope_cos, p.rope_sin, p.rope_pos ? p.rope_pos[(int64_t)b * Tk + key] : key, D, g);
  bf16_t* krow = p.dK + ((int64_t)b * Tk + key) * p.lddk + hk * D;
  bf16_t* vrow = p.dV + ((int64_t)b * Tk + key) * p.lddv + hk * D;
#pragma unroll
  for (int df = 0; df < DF; df++) {
    uint2 w;
    w.x = pack2bf(dk[df][0], dk[df][1]);
    w.y = pack2bf(dk[df][2], dk[df][3]);
    *reinterpret_cast<uint2*>(krow + df * 16 + 4 * g) = w;
    w.x = pack2bf(dv[df][0], dv[df][1]);
    w.y = pack2bf(dv[df][2], dv[df][3]);
    *reinterpret_cast<uint2*>(vrow + df * 16 + 4 * g) = w;
  }
}

// ------------------------------------------------------------------------------------------
// backward dQ, ring form: the arithmetic of attn_bwd_dq_kernel<D, CAUSAL, false, QF> (same fragment mapping, same masks),
// with the 32-key K | V | K^T tiles DMA'd HBM -> LDS into a ring of 3 stages (two tiles in flight while one is consumed),
// counted vmcnt waits and ONE raw barrier per tile (the register-staged form needs two __syncthreads and 6 ds_write_b128
// per thread and tile).  The key-padding mask bytes of a tile ride along as one extra DMA piece of wave 0.
// ------------------------------------------------------------------------------------------
template <int D, bool CAUSAL, int QF, bool PROBE = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_ring_kernel(AttnParams p) {
  constexpr int KD = D / 32;
  constexpr int DF = D / 16;
  constexpr int ROWB = D * 2;
  constexpr int KCH = D / 8;
  constexpr int NS = 3;                       // ring stages
  constexpr int SUB = 32 * ROWB;              // bytes of one sub-tile: [32][D] row-major == [D][32] transposed
  constexpr int STG = 3 * SUB + 1024;         // K | V | K^T | mask[32] (+ the rest of that DMA piece)
  constexpr int NI = SUB / 1024;              // 1 KiB DMA instructions per sub-tile (8 for D = 128, 4 for D = 64)
  constexpr int PS = NI / 4;                  // ... per wave and sub-tile
  constexpr int NU = 3 * PS;                  // tile DMA instructions per wave and stage (wave 0: + 1, the mask line)
  constexpr int RPI = 1024 / ROWB;            // rows of a row-major sub-tile per DMA instruction (4 | 8)
  constexpr int QB = 64 * QF;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  // (giving the four waves the same 32 queries of the four heads of a GQA group instead -- every wave then runs to the same
  // causal diagonal, 19 % fewer workgroup-tiles at the Llama shape -- measured the same kernel time:
  // profiles/r02_attention_bwd.md)
  const AttnBlk blk = attn_blk(p);
  const int b = blk.z, h = blk.y;
  const int hk = h / (p.Hq / p.Hkv);
  const int Tq = p.Tq, Tk = p.Tk, Tqp = p.Tqp, Tkp = p.Tkp;
  const int qb0 = blk.x * QB, qw0 = qb0 + wave * 16 * QF;

  frag_t qf[QF][KD], dof[QF][KD];
  float lse2[QF], delta[QF];
  int qlo[QF];
#pragma unroll
  for (int f = 0; f < QF; f++) {
    const int q = qw0 + f * 16 + li;
    const bool qok = q < Tq;
#pragma unroll
    for (int kd = 0; kd < KD; kd++) {
      qf[f][kd] = qok ? *reinterpret_cast<const frag_t*>(p.Q + ((int64_t)b * Tq + q) * p.ldq + h * D + kd * 32 + g * 8) : zero_frag();
      dof[f][kd] = qok ? *reinterpret_cast<const frag_t*>(p.dO + ((int64_t)b * Tq + q) * p.lddo + h * D + kd * 32 + g * 8) : zero_frag();
    }
    lse2[f] = qok ? p.LSE[((int64_t)b * p.Hq + h) * Tqp + q] * LOG2E : INFINITY;
    float dl = 0.f;   // Delta[q] = sum_d dO[q,d] O[q,d], written out for the dK/dV kernel that runs next on the stream
    if (qok) {
#pragma unroll
      for (int kd = 0; kd < KD; kd++) {
        const frag_t of = *reinterpret_cast<const frag_t*>(p.O + ((int64_t)b * Tq + q) * p.ldo + h * D + kd * 32 + g * 8);
        const u16x8_t ov = __builtin_bit_cast(u16x8_t, of), dv = __builtin_bit_cast(u16x8_t, dof[f][kd]);
#pragma unroll
        for (int e = 0; e < 8; e++) dl = fmaf(bf2f(ov[e]), bf2f(dv[e]), dl);
      }
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    if (qok && g == 0) p.Delta[((int64_t)b * p.Hq + h) * Tqp + q] = dl;
    delta[f] = dl;
    qlo[f] = p.seg_lo ? p.seg_lo[(int64_t)b * Tq + min(q, Tq - 1)] : 0;
  }
  const float sl2 = p.scale * LOG2E;
  // wave-uniform fact for the mask-free path: first key the LAST query of the wave may see (seg_lo is non-decreasing, so
  // every other query of the wave sees at least as far back)
  int qlo_hi = p.seg_lo ? p.seg_lo[(int64_t)b * Tq + min(qw0 + 16 * QF - 1, Tq - 1)] : 0;
  f32x4_t dq[QF][DF];
#pragma unroll
  for (int f = 0; f < QF; f++)
#pragma unroll
    for (int df = 0; df < DF; df++) dq[f][df] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int kend = CAUSAL ? min(Tk, qb0 + QB) : Tk;
  const int ntiles = (kend + 31) / 32;
  const int tbeg = p.seg_lo ? p.seg_lo[(int64_t)b * Tq + min(qb0, Tq - 1)] / 32 : 0;   // packed batches (see AttnParams)
  // every ordinary load above must have RETURNED before the first asm DMA is issued (see attn_bwd_dkdv_ring_kernel)
#pragma unroll
  for (int f = 0; f < QF; f++) {
#pragma unroll
    for (int kd = 0; kd < KD; kd++) asm volatile("" : "+v"(qf[f][kd]), "+v"(dof[f][kd]));
    asm volatile("" : "+v"(lse2[f]), "+v"(delta[f]), "+v"(qlo[f]));
  }
  int tb = tbeg;
  asm volatile("" : "+s"(tb), "+s"(qlo_hi));

  // ---- DMA issue.  Sub-tile sub (0 K, 1 V, 2 K^T) is NI pieces of 1 KiB; this wave owns pieces j = wave + 4 v (v < PS) of
  // each.  Row-major pieces cover RPI rows (lane -> row, 16-byte chunk; the XOR swizzle is applied to the SOURCE chunk, the LDS
  // image is lane-linear), transposed pieces 16 d-rows of 64 bytes.  Rows past the end of the tensor read as zeros (descriptor
  // range check); rows past Tk of a batch in the middle read the next batch's rows, which the key < Tk mask discards. ----
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_offset_of(lds));
  const int nB = p.gz;
  const __amdgpu_buffer_rsrc_t srd_k = __builtin_amdgcn_make_buffer_rsrc((void*)p.K, 0, (unsigned)((((int64_t)nB * Tk - 1) * p.ldk + (int64_t)p.Hkv * D) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_v = __builtin_amdgcn_make_buffer_rsrc((void*)p.V, 0, (unsigned)((((int64_t)nB * Tk - 1) * p.ldv + (int64_t)p.Hkv * D) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_kt = __builtin_amdgcn_make_buffer_rsrc((void*)p.Kt, 0, (unsigned)((int64_t)nB * p.Hkv * D * Tkp * 2), 0x00020000);
  unsigned voff[NU], dsto[NU];
#pragma unroll
  for (int u = 0; u < NU; u++) {
    const int sub = u / PS, j = (u % PS) * 4 + wave;
    if (sub < 2) {
      const int row = j * RPI + lane / KCH, c = lane % KCH;
      const int64_t ld = sub ? p.ldv : p.ldk;
      voff[u] = (unsigned)(((int64_t)b * Tk + row) * ld + hk * D + ((c ^ ring_swz<D>(row)) << 3)) * 2u;
    } else {
      const int d = j * 16 + (lane >> 2), c = lane & 3;
      voff[u] = (unsigned)((((int64_t)b * p.Hkv + hk) * D + d) * Tkp + ((c ^ ((d >> 2) & 3)) << 3)) * 2u;
    }
    dsto[u] = (unsigned)(sub * SUB + j * 1024);
  }
  // the mask line: lanes 0 / 1 carry the 32 mask bytes of the tile, the other lanes repeat lane 0's address
  const uint8_t* mk_src = p.kmask ? p.kmask + (int64_t)b * Tkp + (lane == 1 ? 16 : 0) : reinterpret_cast<const uint8_t*>(p.Kt);
  const unsigned ldk2 = (unsigned)p.ldk * 2u, ldv2 = (unsigned)p.ldv * 2u;
  auto issue = [&](int tile, int s) {
    const int k0 = min(tile, ntiles - 1) * 32;   // past the end: the last tile again, into a stage nobody reads
    const unsigned st = lds0 + (unsigned)(s * STG);
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int sub = u / PS;
      const unsigned dst = __builtin_amdgcn_readfirstlane(st + dsto[u]);
      const unsigned so = sub == 0 ? (unsigned)k0 * ldk2 : (sub == 1 ? (unsigned)k0 * ldv2 : (unsigned)k0 * 2u);
      bufdma16_asm(sub == 0 ? srd_k : (sub == 1 ? srd_v : srd_kt), voff[u] + __builtin_amdgcn_readfirstlane(so), dst);
    }
    if (wave == 0) glds16_asm(mk_src + (p.kmask ? k0 : 0), __builtin_amdgcn_readfirstlane(st + (unsigned)(3 * SUB)));
  };

  // per-lane LDS read offsets (stage 0): K / V fragment kf, A-operand row i = li is tile row 8 (i / 4) + 4 kf + i % 4
  int aA[KD];
  {
    const int row = 8 * (li >> 2) + (li & 3);
#pragma unroll
    for (int kd = 0; kd < KD; kd++) aA[kd] = row * ROWB + (((kd * 4 + g) ^ ring_swz<D>(row)) << 4);
  }
  const int aB = 2 * SUB + li * 64 + ((g ^ ((li >> 2) & 3)) << 4);
  const int aM = 3 * SUB + 8 * g;

  if (ntiles > tb) {
    issue(tb, 0);
    issue(tb + 1, 1);
  }
  int s = 0;
  const bool prb = PROBE && blk.x == 2 && blk.y == 5 && blk.z == 3 && (wave == 0 || wave == 3);
  auto stamp = [&](int it, int i) {
    if constexpr (PROBE) {
      if (prb) {
        const unsigned long long t = __builtin_readcyclecounter();
        if (lane == 0 && it < 16) g_attn_probe[((wave ? 1 : 0) * 16 + it) * 8 + i] = t;
      }
    }
  };
  for (int it = tb; it < ntiles; it++) {
    const int k0 = it * 32;
    stamp(it, 0);
    // tile `it` has landed when at most the younger tile's DMA of this wave is outstanding; after the barrier every wave's
    // share has, and everybody is done reading the stage of tile it - 1, which the next DMA refills
    if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NU + 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NU) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp(it, 1);
    const int sc = s;
    {
      const int s2 = s == 0 ? NS - 1 : s - 1;   // (s + 2) % 3
      issue(it + 2, s2);
    }
    s = s == NS - 1 ? 0 : s + 1;
    stamp(it, 2);
    if (qw0 >= Tq || (CAUSAL && k0 > qw0 + 16 * QF - 1)) continue;

    // LDS reads and their waits placed by hand (see attn_bwd_dkdv_ring_kernel): fragment 0's K / V operands and the mask
    // first, fragment 1's while fragment 0's products run, the K^T operands while fragment 1's run and through the softmax
    const unsigned so = lds0 + (unsigned)(sc * STG);
    unsigned rA[KD];
#pragma unroll
    for (int kd = 0; kd < KD; kd++) rA[kd] = so + (unsigned)aA[kd];
    const unsigned rB = so + (unsigned)aB, rM = so + (unsigned)aM;
    frag_t k0f[KD], v0f[KD], k1f[KD], v1f[KD], ktf[DF];
    u32x2_t mk;
    static_for<0, KD>([&](auto kd) {
      k0f[kd] = lds_read128<0>(rA[kd]);
      v0f[kd] = lds_read128<SUB>(rA[kd]);
    });
    mk = lds_read64<0>(rM);
    f32x4_t st[QF][2], dpt[QF][2];
#pragma unroll
    for (int f = 0; f < QF; f++)
#pragma unroll
      for (int kf = 0; kf < 2; kf++) {
        st[f][kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        dpt[f][kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    static_for<0, KD>([&](auto kd) {
      lds_wait<2 * KD - 1>(k0f[kd], v0f[kd]);
      k1f[kd] = lds_read128<4 * ROWB>(rA[kd]);
      v1f[kd] = lds_read128<SUB + 4 * ROWB>(rA[kd]);
#pragma unroll
      for (int f = 0; f < QF; f++) {
        st[f][0] = mfma16(k0f[kd], qf[f][kd], st[f][0]);
        dpt[f][0] = mfma16(v0f[kd], dof[f][kd], dpt[f][0]);
      }
    });
    static_for<0, KD>([&](auto kd) {
      if constexpr (kd == 0) {
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * KD - 2) : "memory");
        asm volatile("" : "+v"(mk));
        lds_landed(k1f[kd]);
        lds_landed(v1f[kd]);
      } else {
        lds_wait<2 * KD - 2>(k1f[kd], v1f[kd]);
      }
      ktf[2 * kd] = lds_read128<(2 * kd) * 1024>(rB);
      ktf[2 * kd + 1] = lds_read128<(2 * kd + 1) * 1024>(rB);
#pragma unroll
      for (int f = 0; f < QF; f++) {
        st[f][1] = mfma16(k1f[kd], qf[f][kd], st[f][1]);
        dpt[f][1] = mfma16(v1f[kd], dof[f][kd], dpt[f][1]);
      }
    });
    static_assert(DF == 2 * KD, "two K^T fragments are requested per k-step above");
    stamp(it, 3);
    if (!p.kmask) mk = u32x2_t{0x01010101u, 0x01010101u};
    frag_t dsb[QF];
    // interior tile: every (key, query) pair of this wave is visible -- no masks (most tiles of a long causal sequence).
    // Query rows past Tq need no mask: their Q / dO fragments are zero and their LSE is +inf, so P and dS come out as 0.
    const bool all_keys = __all(mk[0] == 0x01010101u && mk[1] == 0x01010101u);
    const bool interior = all_keys && k0 + 32 <= Tk && (!CAUSAL || k0 + 31 <= qw0) && k0 >= qlo_hi;
    if (interior) {
#pragma unroll
      for (int f = 0; f < QF; f++) {
#pragma unroll
        for (int kf = 0; kf < 2; kf++) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float pv = fast_exp2(__builtin_fmaf(st[f][kf][r], sl2, -lse2[f]));
            st[f][kf][r] = pv * (dpt[f][kf][r] - delta[f]) * p.scale;
          }
        }
        dsb[f] = pack_frag(st[f][0], st[f][1]);
      }
    } else {
      // diagonal / boundary / padded tile: query q sees the key RANGE [qlo, min(q, Tk - 1)] (one unsigned compare per element
      // instead of four predicates) minus the padded keys
      const int kbase = k0 + 8 * g;
#pragma unroll
      for (int f = 0; f < QF; f++) {
        const int q = qw0 + f * 16 + li;
        const int hi = q < Tq ? (CAUSAL ? min(q, Tk - 1) : Tk - 1) : qlo[f] - 1;
        const unsigned span = (unsigned)(hi - qlo[f]);
        const bool any = hi >= qlo[f];
        const int rel = kbase - qlo[f];
#pragma unroll
        for (int kf = 0; kf < 2; kf++) {
          const unsigned m4 = kf ? mk[1] : mk[0];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const bool ok = any && (unsigned)(rel + 4 * kf + r) <= span && (m4 & (0xffu << (8 * r))) != 0;
            const float pv = ok ? fast_exp2(__builtin_fmaf(st[f][kf][r], sl2, -lse2[f])) : 0.f;
            st[f][kf][r] = pv * (dpt[f][kf][r] - delta[f]) * p.scale;
          }
        }
        dsb[f] = pack_frag(st[f][0], st[f][1]);
      }
    }
    stamp(it, 4);
    static_for<0, DF>([&](auto df) {
      lds_wait<DF - 1 - df>(ktf[df]);
#pragma unroll
      for (int f = 0; f < QF; f++) dq[f][df] = mfma16(ktf[df], dsb[f], dq[f][df]);
    });
    stamp(it, 5);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail DMAs still target this workgroup's LDS
  int li_e = li;
  asm volatile("" : "+v"(li_e));   // recompute the row addresses here instead of keeping them alive across the tile loop
#pragma unroll
  for (int f = 0; f < QF; f++) {
    const int q = qw0 + f * 16 + li_e;
    if (q >= Tq) continue;
    if (p.rope_cos) rope_grad_inplace<DF>(dq[f], p.rope_cos, p.rope_sin, p.rope_pos ? p.rope_pos[(int64_t)b * Tq + q] : q, D, g);
    bf16_t* orow = p.dQ + ((int64_t)b * Tq + q) * p.lddq + h * D;
#pragma unroll
    for (int df = 0; df < DF; df++) {
      uint2 w;
      w.x = pack2bf(dq[f][df][0], dq[f][df][1]);
      w.y = pack2bf(dq[f][df][2], dq[f][df][3]);
      *reinterpret_cast<uint2*>(orow + df * 16 + 4 * g) = w;
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward dQ, transposed-read form (round 4, shipped): attn_bwd_dq_ring_kernel without the K^T sub-tile.  A stage is K | V | mask
// (17 KiB at D = 128, was 25); the K^T operand of the dQ product is read with ds_read_b64_tr_b16 from the row-major K sub-tile that the S
// product reads with ds_read_b128 (tr_swz keeps both conflict-free).  Same arithmetic / masks / epilogue as the ring kernel.
// ------------------------------------------------------------------------------------------
template <int D, bool CAUSAL, int QF>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_tr_kernel(AttnParams p) {
  constexpr int KD = D / 32;
  constexpr int DF = D / 16;
  constexpr int ROWB = D * 2;
  constexpr int KCH = D / 8;
  constexpr int NS = 3;                       // ring stages
  constexpr int SUB = 32 * ROWB;              // bytes of one row-major sub-tile [32][D]
  constexpr int STG = 2 * SUB + 1024;         // K | V | mask[32] (+ the rest of that DMA piece)
  constexpr int NI = SUB / 1024;              // 1 KiB DMA instructions per sub-tile (8 for D = 128, 4 for D = 64)
  constexpr int PS = NI / 4;                  // ... per wave and sub-tile
  constexpr int NU = 2 * PS;                  // tile DMA instructions per wave and stage (wave 0: + 1, the mask line)
  constexpr int RPI = 1024 / ROWB;            // rows per DMA instruction (4 | 8)
  constexpr int QB = 64 * QF;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const AttnBlk blk = attn_blk(p);
  const int b = blk.z, h = blk.y;
  const int hk = h / (p.Hq / p.Hkv);
  const int Tq = p.Tq, Tk = p.Tk, Tqp = p.Tqp, Tkp = p.Tkp;
  const int qb0 = blk.x * QB, qw0 = qb0 + wave * 16 * QF;

  frag_t qf[QF][KD], dof[QF][KD];
  float lse2[QF], delta[QF];
  int qlo[QF];
#pragma unroll
  for (int f = 0; f < QF; f++) {
    const int q = qw0 + f * 16 + li;
    const bool qok = q < Tq;
#pragma unroll
    for (int kd = 0; kd < KD; kd++) {
      qf[f][kd] = qok ? *reinterpret_cast<const frag_t*>(p.Q + ((int64_t)b * Tq + q) * p.ldq + h * D + kd * 32 + g * 8) : zero_frag();
      dof[f][kd] = qok ? *reinterpret_cast<const frag_t*>(p.dO + ((int64_t)b * Tq + q) * p.lddo + h * D + kd * 32 + g * 8) : zero_frag();
    }
    lse2[f] = qok ? p.LSE[((int64_t)b * p.Hq + h) * Tqp + q] * LOG2E : INFINITY;
    float dl = 0.f;   // Delta[q] = sum_d dO[q,d] O[q,d], written out for the dK/dV kernel that runs next on the stream
    if (qok) {
#pragma unroll
      for (int kd = 0; kd < KD; kd++) {
        const frag_t of = *reinterpret_cast<const frag_t*>(p.O + ((int64_t)b * Tq + q) * p.ldo + h * D + kd * 32 + g * 8);
        const u16x8_t ov = __builtin_bit_cast(u16x8_t, of), dv = __builtin_bit_cast(u16x8_t, dof[f][kd]);
#pragma unroll
        for (int e = 0; e < 8; e++) dl = fmaf(bf2f(ov[e]), bf2f(dv[e]), dl);
      }
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    if (qok && g == 0) p.Delta[((int64_t)b * p.Hq + h) * Tqp + q] = dl;
    delta[f] = dl;
    qlo[f] = p.seg_lo ? p.seg_lo[(int64_t)b * Tq + min(q, Tq - 1)] : 0;
  }
  const float sl2 = p.scale * LOG2E;
  int qlo_hi = p.seg_lo ? p.seg_lo[(int64_t)b * Tq + min(qw0 + 16 * QF - 1, Tq - 1)] : 0;
  f32x4_t dq[QF][DF];
#pragma unroll
  for (int f = 0; f < QF; f++)
#pragma unroll
    for (int df = 0; df < DF; df++) dq[f][df] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int kend = CAUSAL ? min(Tk, qb0 + QB) : Tk;
  const int ntiles = (kend + 31) / 32;
  const int tbeg = p.seg_lo ? p.seg_lo[(int64_t)b * Tq + min(qb0, Tq - 1)] / 32 : 0;
  // every ordinary load above must have RETURNED before the first asm DMA is issued (see attn_bwd_dkdv_ring_kernel)
#pragma unroll
  for (int f = 0; f < QF; f++) {
#pragma unroll
    for (int kd = 0; kd < KD; kd++) asm volatile("" : "+v"(qf[f][kd]), "+v"(dof[f][kd]));
    asm volatile("" : "+v"(lse2[f]), "+v"(delta[f]), "+v"(qlo[f]));
  }
  int tb = tbeg;
  asm volatile("" : "+s"(tb), "+s"(qlo_hi));

  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_offset_of(lds));
  const int nB = p.gz;
  const __amdgpu_buffer_rsrc_t srd_k = __builtin_amdgcn_make_buffer_rsrc((void*)p.K, 0, (unsigned)((((int64_t)nB * Tk - 1) * p.ldk + (int64_t)p.Hkv * D) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_v = __builtin_amdgcn_make_buffer_rsrc((void*)p.V, 0, (unsigned)((((int64_t)nB * Tk - 1) * p.ldv + (int64_t)p.Hkv * D) * 2), 0x00020000);
  unsigned voff[NU], dsto[NU];
#pragma unroll
  for (int u = 0; u < NU; u++) {
    const int sub = u / PS, j = (u % PS) * 4 + wave;
    const int row = j * RPI + lane / KCH, c = lane % KCH;
    const int64_t ld = sub ? p.ldv : p.ldk;
    voff[u] = (unsigned)(((int64_t)b * Tk + row) * ld + hk * D + ((c ^ tr_swz<D>(row)) << 3)) * 2u;
    dsto[u] = (unsigned)(sub * SUB + j * 1024);
  }
  const uint8_t* mk_src = p.kmask ? p.kmask + (int64_t)b * Tkp + (lane == 1 ? 16 : 0) : reinterpret_cast<const uint8_t*>(p.K);
  const unsigned ldk2 = (unsigned)p.ldk * 2u, ldv2 = (unsigned)p.ldv * 2u;
  auto issue = [&](int tile, int s) {
    const int k0 = min(tile, ntiles - 1) * 32;   // past the end: the last tile again, into a stage nobody reads
    const unsigned st = lds0 + (unsigned)(s * STG);
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int sub = u / PS;
      const unsigned dst = __builtin_amdgcn_readfirstlane(st + dsto[u]);
      const unsigned so = sub == 0 ? (unsigned)k0 * ldk2 : (unsigned)k0 * ldv2;
      bufdma16_asm(sub == 0 ? srd_k : srd_v, voff[u] + __builtin_amdgcn_readfirstlane(so), dst);
    }
    if (wave == 0) glds16_asm(mk_src + (p.kmask ? k0 : 0), __builtin_amdgcn_readfirstlane(st + (unsigned)(2 * SUB)));
  };

  int aA[KD];
  {
    const int row = 8 * (li >> 2) + (li & 3);
#pragma unroll
    for (int kd = 0; kd < KD; kd++) aA[kd] = row * ROWB + (((kd * 4 + g) ^ tr_swz<D>(row)) << 4);
  }
  const int aT0 = (int)tr_lane_off<D>(g, li);
  const int aM = 2 * SUB + 8 * g;

  if (ntiles > tb) {
    issue(tb, 0);
    issue(tb + 1, 1);
  }
  int s = 0;
  constexpr int NPAIR = (DF >= 8) ? 3 : KD;   // k-steps of the fragment-1 products that each request two transposed fragments (4 reads)
  constexpr int AH = 2 * NPAIR;               // transposed fragments in flight through the softmax (12 reads at D = 128: the LDS counter holds 15)
  for (int it = tb; it < ntiles; it++) {
    const int k0 = it * 32;
    if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NU + 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NU) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int sc = s;
    {
      const int s2 = s == 0 ? NS - 1 : s - 1;   // (s + 2) % 3
      issue(it + 2, s2);
    }
    s = s == NS - 1 ? 0 : s + 1;
    if (qw0 >= Tq || (CAUSAL && k0 > qw0 + 16 * QF - 1)) continue;

    const unsigned so = lds0 + (unsigned)(sc * STG);
    unsigned rA[KD];
#pragma unroll
    for (int kd = 0; kd < KD; kd++) rA[kd] = so + (unsigned)aA[kd];
    const unsigned rM = so + (unsigned)aM;
    unsigned rT[DF];
#pragma unroll
    for (int df = 0; df < DF; df++) rT[df] = (so + (unsigned)aT0) ^ (unsigned)(df << 5);
    frag_t k0f[KD], v0f[KD], k1f[KD], v1f[KD];
    TrFrag ktf[DF];
    u32x2_t mk;
    static_for<0, KD>([&](auto kd) {
      k0f[kd] = lds_read128<0>(rA[kd]);
      v0f[kd] = lds_read128<SUB>(rA[kd]);
    });
    mk = lds_read64<0>(rM);
    f32x4_t st[QF][2], dpt[QF][2];
#pragma unroll
    for (int f = 0; f < QF; f++)
#pragma unroll
      for (int kf = 0; kf < 2; kf++) {
        st[f][kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        dpt[f][kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    static_for<0, KD>([&](auto kd) {
      lds_wait<2 * KD - 1>(k0f[kd], v0f[kd]);
      k1f[kd] = lds_read128<4 * ROWB>(rA[kd]);
      v1f[kd] = lds_read128<SUB + 4 * ROWB>(rA[kd]);
#pragma unroll
      for (int f = 0; f < QF; f++) {
        st[f][0] = mfma16(k0f[kd], qf[f][kd], st[f][0]);
        dpt[f][0] = mfma16(v0f[kd], dof[f][kd], dpt[f][0]);
      }
    });
    static_for<0, KD>([&](auto kd) {
      constexpr int younger = 2 * (KD - 1 - kd) + 4 * (kd < NPAIR ? kd : NPAIR);
      if constexpr (kd == 0) {
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger) : "memory");
        asm volatile("" : "+v"(mk));
        lds_landed(k1f[kd]);
        lds_landed(v1f[kd]);
      } else {
        lds_wait<younger>(k1f[kd], v1f[kd]);
      }
      if constexpr (kd < NPAIR) {
        ktf[2 * kd].lo = lds_read_tr<0>(rT[2 * kd]);
        ktf[2 * kd].hi = lds_read_tr<4 * ROWB>(rT[2 * kd]);
        ktf[2 * kd + 1].lo = lds_read_tr<0>(rT[2 * kd + 1]);
        ktf[2 * kd + 1].hi = lds_read_tr<4 * ROWB>(rT[2 * kd + 1]);
      }
#pragma unroll
      for (int f = 0; f < QF; f++) {
        st[f][1] = mfma16(k1f[kd], qf[f][kd], st[f][1]);
        dpt[f][1] = mfma16(v1f[kd], dof[f][kd], dpt[f][1]);
      }
    });
    static_assert(DF == 2 * KD && AH <= DF, "two K^T fragments are requested per k-step above");
    if (!p.kmask) mk = u32x2_t{0x01010101u, 0x01010101u};
    frag_t dsb[QF];
    const bool all_keys = __all(mk[0] == 0x01010101u && mk[1] == 0x01010101u);
    const bool interior = all_keys && k0 + 32 <= Tk && (!CAUSAL || k0 + 31 <= qw0) && k0 >= qlo_hi;
    if (interior) {
#pragma unroll
      for (int f = 0; f < QF; f++) {
#pragma unroll
        for (int kf = 0; kf < 2; kf++) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float pv = fast_exp2(__builtin_fmaf(st[f][kf][r], sl2, -lse2[f]));
            st[f][kf][r] = pv * (dpt[f][kf][r] - delta[f]) * p.scale;
          }
        }
        dsb[f] = pack_frag(st[f][0], st[f][1]);
      }
    } else {
      const int kbase = k0 + 8 * g;
#pragma unroll
      for (int f = 0; f < QF; f++) {
        const int q = qw0 + f * 16 + li;
        const int hi = q < Tq ? (CAUSAL ? min(q, Tk - 1) : Tk - 1) : qlo[f] - 1;
        const unsigned span = (unsigned)(hi - qlo[f]);
        const bool any = hi >= qlo[f];
        const int rel = kbase - qlo[f];
#pragma unroll
        for (int kf = 0; kf < 2; kf++) {
          const unsigned m4 = kf ? mk[1] : mk[0];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const bool ok = any && (unsigned)(rel + 4 * kf + r) <= span && (m4 & (0xffu << (8 * r))) != 0;
            const float pv = ok ? fast_exp2(__builtin_fmaf(st[f][kf][r], sl2, -lse2[f])) : 0.f;
            st[f][kf][r] = pv * (dpt[f][kf][r] - delta[f]) * p.scale;
          }
        }
        dsb[f] = pack_frag(st[f][0], st[f][1]);
      }
    }
    static_for<0, DF>([&](auto df) {
      constexpr int inflight = (DF - df < AH ? DF - df : AH);   // fragments requested and not yet waited for, incl. df
      lds_wait<2 * (inflight - 1)>(ktf[df]);
      if constexpr (df + AH < DF) {
        ktf[df + AH].lo = lds_read_tr<0>(rT[df + AH]);
        ktf[df + AH].hi = lds_read_tr<4 * ROWB>(rT[df + AH]);
      }
      const frag_t a = tr_join(ktf[df]);
#pragma unroll
      for (int f = 0; f < QF; f++) dq[f][df] = mfma16(a, dsb[f], dq[f][df]);
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail DMAs still target this workgroup's LDS
  int li_e = li;
  asm volatile("" : "+v"(li_e));
#pragma unroll
  for (int f = 0; f < QF; f++) {
    const int q = qw0 + f * 16 + li_e;
    if (q >= Tq) continue;
    if (p.rope_cos) rope_grad_inplace<DF>(dq[f], p.rope_cos, p.rope_sin, p.rope_pos ? p.rope_pos[(int64_t)b * Tq + q] : q, D, g);
    bf16_t* orow = p.dQ + ((int64_t)b * Tq + q) * p.lddq + h * D;
#pragma unroll
    for (int df = 0; df < DF; df++) {
      uint2 w;
      w.x = pack2bf(dq[f][df][0], dq[f][df][1]);
      w.y = pack2bf(dq[f][df][2], dq[f][df][3]);
      *reinterpret_cast<uint2*>(orow + df * 16 + 4 * g) = w;
    }
  }
}

// ---- gradients of WavLM's gated relative position bias from dL/d(score) (written by attn_bwd_dq_kernel<..., RP>) ----
// score(b, h, q, k) += gate[b, h, q] * tab[h][k - q + T - 1]  (src/slam_llm/models/wavlm/modules.py:504-533)
//   d gate[b, h, q] = sum_k ds[b, h, q, k] * tab[h][k - q + T - 1]                      one wave per row, fixed-order lane sums
//   d tab[h][r]    += sum_b sum_q gate[b, h, q] * ds[b, h, q, q + r - (T - 1)]           one thread per distance r: consecutive threads
//                                                                                        read consecutive keys; no atomics, bit-reproducible
__global__ __launch_bounds__(256) void relpos_dgate_kernel(const float* __restrict__ ds, const float* __restrict__ tab, float* __restrict__ dgate,
                                                           int B, int H, int Tq, int Tk, int Tkp, int Tqp, int rp_T, int rp_ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t rows = (int64_t)B * H * Tq;
  for (int64_t row = blockIdx.x * 4ll + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const int q = (int)(row % Tq);
    const int64_t bh = row / Tq;
    const int h = (int)(bh % H);
    const float* d = ds + row * Tkp;
    const float* t = tab + (int64_t)h * rp_ld + (rp_T - 1 - q);
    float acc = 0.f;
    for (int k = lane; k < Tk; k += 64) acc = fmaf(d[k], t[k], acc);
    acc = wave_sum(acc);
    if (lane == 0) dgate[bh * Tqp + q] = acc;
  }
}

__global__ __launch_bounds__(256) void relpos_dtab_kernel(const float* __restrict__ ds, const float* __restrict__ gate, float* __restrict__ dtab,
                                                          int B, int H, int Tq, int Tk, int Tkp, int Tqp, int rp_T, int rp_ld) {
  const int r = blockIdx.x * 256 + threadIdx.x;     // table index = k - q + rp_T - 1
  const int h = blockIdx.y;
  if (r >= 2 * rp_T - 1) return;
  float acc = 0.f;
  for (int b = 0; b < B; b++) {
    const float* d = ds + ((int64_t)b * H + h) * Tq * Tkp;
    const float* gt = gate + ((int64_t)b * H + h) * Tqp;
    for (int q = 0; q < Tq; q++) {
      const int k = q + r - (rp_T - 1);
      if (k >= 0 && k < Tk) acc = fmaf(gt[q], d[(int64_t)q * Tkp + k], acc);
    }
  }
  dtab[(int64_t)h * rp_ld + r] += acc;
}

int g_attn_tr = 1;    // 1 = transposed operands by ds_read_b64_tr_b16 from the row-major tiles (shipped), 0 = round-3 kernels on the [B,H,D,Tp] copies
int g_attn_xcd = 1;   // 1 = XCD-aware workgroup numbering (shipped), 0 = hardware round-robin order (A/B in tools)
int g_attn_heavy = 1; // 1 = causal launches start each XCD's heaviest sequence blocks first (round 5), 0 = id order (A/B: slam_attn_set_fwd_qf 50 / 51)

// every attention kernel is launched through this: logical 3-D grid -> 1-D launch + the geometry attn_blk() needs
template <class Kern>
static void attn_launch(Kern kern, dim3 grid, unsigned threads, int lds, hipStream_t s, AttnParams p, int heavy = 0) {
  p.gx = (int)grid.x; p.gy = (int)grid.y; p.gz = (int)grid.z;
  p.xcd = g_attn_xcd;
  p.heavy = (g_attn_heavy && !p.seg_lo) ? heavy : 0;   // (packed batches: a block's work follows its segment, not its index)
  hipLaunchKernelGGL(kern, dim3(grid.x * grid.y * grid.z), dim3(threads), lds, s, p);
}

template <int D, bool CAUSAL, int QF, bool PROBE = false>
int launch_dq_ring(const AttnParams& p, dim3 grid, hipStream_t s) {
  constexpr int lds = 3 * (3 * 32 * D * 2 + 1024);
  static bool attr_set = false;
  auto kern = attn_bwd_dq_ring_kernel<D, CAUSAL, QF, PROBE>;
  constexpr int heavy = CAUSAL ? -1 : 0;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      slam_set_error("slam_attn_bwd: cannot raise the LDS limit to %d", lds);
      return -2;
    }
    attr_set = true;
  }
  attn_launch(kern, grid, 256, lds, s, p, heavy);
  return 0;
}

template <int D, bool CAUSAL, int ABL = 0>
int launch_dkdv_ring(const AttnParams& p, dim3 grid, hipStream_t s) {
  constexpr int lds = 4 * (4 * 32 * D * 2 + 1024);
  static bool attr_set = false;
  auto kern = attn_bwd_dkdv_ring_kernel<D, CAUSAL, ABL>;
  constexpr int heavy = CAUSAL ? 1 : 0;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      slam_set_error("slam_attn_bwd: cannot raise the LDS limit to %d", lds);
      return -2;
    }
    attr_set = true;
  }
  attn_launch(kern, grid, 512, lds, s, p, heavy);
  return 0;
}

template <int D, bool CAUSAL, int QF>
int launch_dq_tr(const AttnParams& p, dim3 grid, hipStream_t s) {
  constexpr int lds = 3 * (2 * 32 * D * 2 + 1024);
  static bool attr_set = false;
  auto kern = attn_bwd_dq_tr_kernel<D, CAUSAL, QF>;
  constexpr int heavy = CAUSAL ? -1 : 0;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      slam_set_error("slam_attn_bwd: cannot raise the LDS limit to %d", lds);
      return -2;
    }
    attr_set = true;
  }
  attn_launch(kern, grid, 256, lds, s, p, heavy);
  return 0;
}

}  // namespace
// (round 5's 4-wave x 32-key form of the kernel below -- half the LDS bytes per MFMA, bit-identical, 11 % slower, profiles/r05_attention_dkdv32.md --
// was removed in round 6 with its translation unit; git history keeps it)
namespace {

template <int D, bool CAUSAL>
int launch_dkdv_tr(const AttnParams& p, dim3 grid, hipStream_t s) {
  constexpr int lds = 4 * (2 * 32 * D * 2 + 1024);
  static bool attr_set = false;
  auto kern = attn_bwd_dkdv_tr_kernel<D, CAUSAL>;
  constexpr int heavy = CAUSAL ? 1 : 0;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      slam_set_error("slam_attn_bwd: cannot raise the LDS limit to %d", lds);
      return -2;
    }
    attr_set = true;
  }
  attn_launch(kern, grid, 512, lds, s, p, heavy);
  return 0;
}

int g_attn_bwd_variant = 0;   // 0 = ring kernels (shipped), 1 = round-1 kernels, 2 = ring dK/dV + register-staged dQ (A/B in tools)
int g_attn_fwd_qf = 0;   // 16-row query fragments per wave of the forward kernel: 0 = auto, 1 / 2 forced (tools)

int check_common(const char* name, int64_t B, int64_t Tq, int64_t Tk, int64_t Tqp, int64_t Tkp, int64_t Hq, int64_t Hkv,
                 int64_t D, int causal) {
  SLAM_CHECK_ARG(D == 64 || D == 128, "%s: head_dim %ld unsupported (64|128)", name, (long)D);
  SLAM_CHECK_ARG(B > 0 && Tq > 0 && Tk > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "%s: bad shape B=%ld Tq=%ld Tk=%ld Hq=%ld Hkv=%ld",
                 name, (long)B, (long)Tq, (long)Tk, (long)Hq, (long)Hkv);
  SLAM_CHECK_ARG(Tqp % 64 == 0 && Tqp >= Tq && Tkp % 64 == 0 && Tkp >= Tk, "%s: padded lengths must be multiples of 64 and >= T", name);
  SLAM_CHECK_ARG(!causal || Tq == Tk, "%s: causal attention needs Tq == Tk", name);
  SLAM_CHECK_ARG(B < 65536 && Hq < 65536, "%s: grid dims exceed 65535", name);
  return 0;
}

}  // namespace

extern "C" int slam_attn_debug_clock(unsigned long long* out256) {   // tools: stamps of the last PROBE dQ launch (variant 14; sync first)
  SLAM_CHECK_ARG(out256 != nullptr, "slam_attn_debug_clock: null output");
  hipError_t e = hipMemcpyFromSymbol(out256, HIP_SYMBOL(g_attn_probe), 256 * sizeof(unsigned long long));
  if (e != hipSuccess) {
    slam_set_error("slam_attn_debug_clock: %s", hipGetErrorString(e));
    return -2;
  }
  return 0;
}

extern "C" int slam_attn_set_bwd_variant(int variant) {   // tools: 0 = DMA-ring backward kernels (shipped), 1 = round-1 kernels
  SLAM_CHECK_ARG(variant == 0 || variant == 1 || variant == 2 || variant == 14 || variant == 11 || variant == 12 || variant == 15 || variant == 19, "slam_attn_set_bwd_variant: %d (0 | 1 | 2 | 11, 12, 15 timing ablations)", variant);
  g_attn_bwd_variant = variant;
  return 0;
}

extern int g_attn_fwd_dma, g_attn_fwd_plain, g_attn_fwd_qs;
extern "C" int slam_attn_set_fwd_qf(int qf) {   // tools: 0 = auto, 1 / 2 fragments per wave; 10 / 11 = register-staged / DMA tiles
  SLAM_CHECK_ARG((qf >= 0 && qf <= 2) || qf == 10 || qf == 11 || qf == 20 || qf == 21 || qf == 30 || qf == 31 || qf == 40 || qf == 41 || qf == 50 || qf == 51 || qf == 60 || qf == 61,
                 "slam_attn_set_fwd_qf: %d (0 = auto, 1 or 2; 10 / 11 = staged / DMA tiles; 20 / 21 = hardware / XCD-aware workgroup order; "
                 "30 / 31 = general / mask-free instantiation for unmasked bidirectional D = 64; 40 / 41 = transposed-copy / transposed-read kernels; "
                 "50 / 51 = id order / heaviest sequence block first in causal launches; 60 / 61 = general softmax / Q pre-scaled by the caller "
                 "and accumulators started at -m in LSE-less mask-free launches)", qf);
  if (qf >= 60) g_attn_fwd_qs = qf - 60;
  else if (qf >= 50) g_attn_heavy = qf - 50;
  else if (qf >= 40) g_attn_tr = qf - 40;
  else if (qf >= 30) g_attn_fwd_plain = qf - 30;
  else if (qf >= 20) g_attn_xcd = qf - 20;   // (all attention kernels, forward and backward)
  else if (qf >= 10) g_attn_fwd_dma = qf - 10;
  else g_attn_fwd_qf = qf;
  return 0;
}

// every attention tuning knob back to the value it is DEFINED with (slam_reset_tuning)
void slam_attn_reset_tuning_() {
  g_attn_tr = 1; g_attn_xcd = 1; g_attn_heavy = 1; g_attn_bwd_variant = 0; g_attn_fwd_qf = 0;
  g_attn_fwd_qs = 1; g_attn_fwd_plain = 1; g_attn_fwd_dma = 1;
}

int g_attn_fwd_qs = 1;    // 1 = LSE-less launches of the mask-free form whose Q arrives pre-scaled (negative scale) run attn_fwd_kernel<..., QS>
int g_attn_fwd_plain = 1; // 1 = unmasked bidirectional D = 64 launches (Whisper) take the instantiation without mask / segment bookkeeping
int g_attn_fwd_dma = 1;   // 1 = K / V^T tiles by LDS-DMA ring (shipped), 0 = register-staged tiles (A/B in tools)

// forward launch: the DMA form whenever the 32-bit byte offsets of its descriptors can address K and V^T
template <int D, bool CAUSAL, int QF, bool RP = false, bool DROP = false, bool PROBE = false>
static void launch_fwd(const AttnParams& p, int64_t B, hipStream_t s) {
  dim3 grid((unsigned)cdiv64(p.Tq, 64 * QF), (unsigned)p.Hq, (unsigned)B);
  const int64_t lim = (int64_t)1 << 31;
  // row-major V whenever the caller passed it (and the [B,H,D,Tp] copy only when it did not, or for the A/B knob)
  const bool trv = p.V != nullptr && (g_attn_tr || p.Vt == nullptr);
  const bool fits = (B * p.Tk * p.ldk + (int64_t)p.Hkv * D) * 2 < lim &&
                    (trv ? (B * p.Tk * p.ldv + (int64_t)p.Hkv * D) * 2 < lim : B * p.Hkv * D * p.Tkp * 2 < lim);
  if constexpr (!PROBE) {
    if (trv) {
      if constexpr (D == 64 && !CAUSAL && QF == 2 && !RP && !DROP) {
        if (fits && g_attn_fwd_dma && !p.kmask && !p.seg_lo && g_attn_fwd_plain) {
          // Q pre-scaled by the caller and no LSE wanted (the frozen encoder's forward): accumulators start at -m
          if (!p.LSE && p.qpre && g_attn_fwd_qs) attn_launch((attn_fwd_kernel<D, CAUSAL, QF, RP, DROP, PROBE, true, true, true, true>), grid, 256, 0, s, p, 0);
          else attn_launch((attn_fwd_kernel<D, CAUSAL, QF, RP, DROP, PROBE, true, true, true>), grid, 256, 0, s, p, CAUSAL ? -1 : 0);
          return;
        }
      }
      if (fits && g_attn_fwd_dma) attn_launch((attn_fwd_kernel<D, CAUSAL, QF, RP, DROP, PROBE, true, false, true>), grid, 256, 0, s, p, CAUSAL ? -1 : 0);
      else attn_launch((attn_fwd_kernel<D, CAUSAL, QF, RP, DROP, PROBE, false, false, true>), grid, 256, 0, s, p, CAUSAL ? -1 : 0);
      return;
    }
  }
  if constexpr (D == 64 && !CAUSAL && QF == 2 && !RP && !DROP && !PROBE) {
    if (fits && g_attn_fwd_dma && !p.kmask && !p.seg_lo && g_attn_fwd_plain) {
      attn_launch((attn_fwd_kernel<D, CAUSAL, QF, RP, DROP, PROBE, true, true>), grid, 256, 0, s, p, CAUSAL ? -1 : 0);
      return;
    }
  }
  if (fits && g_attn_fwd_dma) attn_launch((attn_fwd_kernel<D, CAUSAL, QF, RP, DROP, PROBE, true>), grid, 256, 0, s, p, CAUSAL ? -1 : 0);
  else attn_launch((attn_fwd_kernel<D, CAUSAL, QF, RP, DROP, PROBE, false>), grid, 256, 0, s, p, CAUSAL ? -1 : 0);
}

// what the 32-bit byte offsets of the DMA descriptors can address: the row-major operands (Q / dO and K / V live inside fused buffers:
// ld = 3 H D for a QKV buffer, so the range is exhausted ~3x sooner than H D would suggest), and the [B,H,D,Tp] copies of the round-3
// ring kernels.  ONE predicate for slam_attn_needs_transposed and for every backward launch (they used to differ, and the
// dK / dV launch did not look at all -- beyond 2 GiB its descriptor reads would have returned zeros, silently).
struct AttnFits { bool rowmajor, copies; };
static AttnFits attn_fits(int64_t B, int64_t Tq, int64_t Tk, int64_t Tqp, int64_t Tkp, int64_t Hq, int64_t Hkv, int64_t D, int64_t ldq,
                          int64_t ldk, int64_t ldv, int64_t lddo) {
  const int64_t lim = (int64_t)1 << 31;
  AttnFits f;
  f.rowmajor = (B * Tk * ldk + Hkv * D) * 2 < lim && (B * Tk * ldv + Hkv * D) * 2 < lim &&
               (B * Tq * ldq + Hq * D) * 2 < lim && (B * Tq * lddo + Hq * D) * 2 < lim;
  f.copies = B * Hkv * D * Tkp * 2 < lim && B * Hq * D * Tqp * 2 < lim;
  return f;
}

// which transposed [B,H,D,Tp] copies a call with these arguments reads (bit 0: slam_attn_fwd needs Vt; bit 1: slam_attn_bwd needs Qt / Kt /
// dOt).  0 for everything the transposed-read kernels cover; the host side builds the copies (slam_head_rope_transpose) only when asked
// to.  flags: bit 0 = attention-probability dropout, bit 1 = gated relative position bias (WavLM).
extern "C" int slam_attn_needs_transposed(int64_t B, int64_t Tq, int64_t Tk, int64_t Hq, int64_t Hkv, int64_t D, int64_t ldq, int64_t ldk,
                                          int64_t ldv, int64_t lddo, int flags) {
  int need = 0;
  if (!g_attn_tr) return 3;
  // backward: the transposed-read kernels are the ring forms; <= 64 queries, dropout, the relative position bias, the non-default
  // variants and tensors beyond the descriptors' 2 GiB stay on the register-staged kernels, which read the copies
  if (flags != 0 || Tq <= 64 || !attn_fits(B, Tq, Tk, 0, 0, Hq, Hkv, D, ldq, ldk, ldv, lddo).rowmajor || g_attn_bwd_variant != 0) need |= 2;
  return need;   // (the forward kernel has a transposed-read form of every instantiation, register-staged ones included)
}

extern "C" int slam_attn_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* Vt, const void* V, int64_t ldv,
                             void* O, int64_t ldo, float* LSE, const uint8_t* key_mask, int64_t B,
                             int64_t Tq, int64_t Tk, int64_t Tqp, int64_t Tkp, int64_t Hq, int64_t Hkv, int64_t D,
                             int causal, float scale, const int32_t* seg_lo, const int32_t* seg_hi, const float* rp_gate,
                             const float* rp_tab, int64_t rp_T, int64_t rp_ld, float drop_p, uint64_t drop_seed, void* stream) {
  SLAM_CHECK_ARG(Q && K && (Vt || V) && O, "slam_attn_fwd: null pointer (V row-major and / or its [B,H,D,Tp] copy Vt must be given)");
  SLAM_CHECK_ARG(!V || ldv % 8 == 0, "slam_attn_fwd: ldv must be a multiple of 8");
  SLAM_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "slam_attn_fwd: drop_p=%f must be in [0, 1)", (double)drop_p);
  SLAM_CHECK_ARG(drop_p == 0.f || (D == 64 && !causal && !seg_lo),
                 "slam_attn_fwd: attention-probability dropout is implemented for head_dim 64, bidirectional, unpacked batches");
  SLAM_CHECK_ARG((rp_gate == nullptr) == (rp_tab == nullptr), "slam_attn_fwd: rp_gate / rp_tab must both be set or both null");
  SLAM_CHECK_ARG(!rp_gate || (D == 64 && !causal && !seg_lo && rp_T >= Tq && rp_T >= Tk && rp_ld >= 2 * rp_T - 1),
                 "slam_attn_fwd: the gated relative position bias is implemented for head_dim 64, bidirectional, unpacked batches, "
                 "with a table covering the sequence (rp_T >= T, rp_ld >= 2 rp_T - 1)");
  SLAM_CHECK_ARG(!seg_lo || Tq == Tk, "slam_attn_fwd: packed sequences (seg_lo) need self-attention");
  SLAM_CHECK_ARG(!seg_lo || causal || seg_hi, "slam_attn_fwd: bidirectional packed sequences need seg_hi (per query row)");
  if (int rc = check_common("slam_attn_fwd", B, Tq, Tk, Tqp, Tkp, Hq, Hkv, D, causal)) return rc;
  SLAM_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 4 == 0, "slam_attn_fwd: leading dims must be multiples of 8");
  AttnParams p = {};
  p.Q = (const bf16_t*)Q; p.ldq = ldq; p.K = (const bf16_t*)K; p.ldk = ldk; p.Vt = (const bf16_t*)Vt;
  p.V = (const bf16_t*)V; p.ldv = ldv;
  p.O = (bf16_t*)O; p.ldo = ldo; p.LSE = LSE; p.kmask = key_mask;
  p.Tq = (int)Tq; p.Tk = (int)Tk; p.Tqp = (int)Tqp; p.Tkp = (int)Tkp; p.Hq = (int)Hq; p.Hkv = (int)Hkv;
  // scale < 0: Q was produced multiplied by |scale| * log2(e) (a frozen query projection with the factor folded into its weights)
  p.qpre = scale < 0.f ? 1 : 0;
  p.scale = scale < 0.f ? -scale : scale;
  SLAM_CHECK_ARG(!p.qpre || (!rp_gate && drop_p == 0.f), "slam_attn_fwd: a pre-scaled Q (negative scale) is for the plain forward (no relative position bias, no dropout)");
  p.seg_lo = seg_lo;
  p.seg_hi = causal ? nullptr : seg_hi;   // (the causal forward only needs the sequence starts)
  p.rp_gate = rp_gate; p.rp_tab = rp_tab; p.rp_T = (int)rp_T; p.rp_ld = (int)rp_ld;
  p.drop_thresh = drop_p > 0.f ? slam_drop_thresh16(drop_p) : 0u;
  p.drop_scale = 1.0f / (1.0f - drop_p);
  p.drop_seed = drop_seed;
  p.drop_salt = g_slam_drop_salt;
  hipStream_t s = (hipStream_t)stream;
  if (drop_p > 0.f && rp_gate) {   // un-frozen WavLM in train mode: attention_dropout on top of the gated bias (WavLM.py:181, modules.py:504-562)
    launch_fwd<64, false, 2, true, true>(p, B, s);
  } else if (drop_p > 0.f) {
    launch_fwd<64, false, 2, false, true>(p, B, s);
  } else if (rp_gate) {
    launch_fwd<64, false, 2, true>(p, B, s);
  } else if (g_attn_bwd_variant == 14 && D == 64 && !causal) {   // tools/attn_fwd_probe.py: the same kernel with cycle stamps
    launch_fwd<64, false, 2, false, false, true>(p, B, s);
  } else {
    // measured (tools/attn_bwd_bench.py, interleaved): with LDS-DMA tiles two fragments per wave win at both head sizes (D = 128,
    // Llama shape: 86 vs 108 us -- half the DMA pieces per query, and no staging registers push it past 256 VGPRs any more); with
    // register-staged tiles (tensors beyond the descriptors' 2 GiB) D = 128 prefers one fragment (119 vs 134 us: 261 VGPRs = one
    // wave per SIMD with two)
    const int64_t lim = (int64_t)1 << 31;
    const bool dma = g_attn_fwd_dma && (B * Tk * ldk + Hkv * D) * 2 < lim && B * Hkv * D * Tkp * 2 < lim;
    const int qf = g_attn_fwd_qf ? g_attn_fwd_qf : ((D == 128 && !dma) ? 1 : 2);
    if (D == 64) {
      if (causal) { if (qf == 2) launch_fwd<64, true, 2>(p, B, s); else launch_fwd<64, true, 1>(p, B, s); }
      else { if (qf == 2) launch_fwd<64, false, 2>(p, B, s); else launch_fwd<64, false, 1>(p, B, s); }
    } else {
      if (causal) { if (qf == 2) launch_fwd<128, true, 2>(p, B, s); else launch_fwd<128, true, 1>(p, B, s); }
      else { if (qf == 2) launch_fwd<128, false, 2>(p, B, s); else launch_fwd<128, false, 1>(p, B, s); }
    }
  }
  SLAM_CHECK_LAUNCH("slam_attn_fwd");
  return 0;
}

// dQ launch: 32 queries per wave (128 per workgroup) once there is more than one 64-query block; the descriptor (DMA ring) forms
// whenever attn_fits says their descriptors reach (variant 2: register-staged form, for A/B in tools)
template <int D, bool CAUSAL>
static int launch_dq(const AttnParams& p, int64_t B, hipStream_t s, AttnFits fits) {
  if (p.Tq > 64 && g_attn_bwd_variant != 1) {
    dim3 g2((unsigned)cdiv64(p.Tq, 128), (unsigned)p.Hq, (unsigned)B);
    if (fits.rowmajor && fits.copies && g_attn_bwd_variant == 14) {
      if constexpr (D == 128 && CAUSAL) return launch_dq_ring<D, CAUSAL, 2, true>(p, g2, s);
    }
    if (g_attn_bwd_variant != 2) {
      if (g_attn_tr && fits.rowmajor) return launch_dq_tr<D, CAUSAL, 2>(p, g2, s);
      if (!g_attn_tr && fits.rowmajor && fits.copies) return launch_dq_ring<D, CAUSAL, 2>(p, g2, s);
    }
    attn_launch((attn_bwd_dq_kernel<D, CAUSAL, false, 2>), g2, 256, 0, s, p, CAUSAL ? -1 : 0);      // (reads Kt: slam_attn_needs_transposed asked for it)
  } else {
    dim3 g1((unsigned)cdiv64(p.Tq, 64), (unsigned)p.Hq, (unsigned)B);
    attn_launch((attn_bwd_dq_kernel<D, CAUSAL, false, 1>), g1, 256, 0, s, p, CAUSAL ? -1 : 0);
  }
  return 0;
}

// dK / dV launch: transposing-read ring kernel, round-3 ring kernel on the copies (tools: slam_attn_set_fwd_qf 40), or the register-staged
// round-1 kernel (variant 1, and every tensor the descriptors cannot address)
template <int D, bool CAUSAL>
static int launch_dkdv(const AttnParams& p, int64_t B, hipStream_t s, AttnFits fits) {
  dim3 gk((unsigned)cdiv64(p.Tk, 64), (unsigned)p.Hkv, (unsigned)B);
  dim3 gk2((unsigned)cdiv64(p.Tk, 128), (unsigned)p.Hkv, (unsigned)B);
  if (g_attn_bwd_variant != 1) {
    if (g_attn_tr && fits.rowmajor) return launch_dkdv_tr<D, CAUSAL>(p, gk2, s);
    if (!g_attn_tr && fits.rowmajor && fits.copies) return launch_dkdv_ring<D, CAUSAL>(p, gk2, s);
  }
  attn_launch((attn_bwd_dkdv_kernel<D, CAUSAL>), gk, 256, 0, s, p, CAUSAL ? 1 : 0);
  return 0;
}

extern "C" int slam_attn_bwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V,
                             int64_t ldv, const void* Qt, const void* Kt, const void* O, int64_t ldo,
                             const void* dO, int64_t lddo, const void* dOt, const float* LSE,
                             float* Delta, const uint8_t* key_mask, void* dQ, int64_t lddq, void* dK,
                             int64_t lddk, void* dV, int64_t lddv, int64_t B, int64_t Tq, int64_t Tk, int64_t Tqp,
                             int64_t Tkp, int64_t Hq, int64_t Hkv, int64_t D, int causal, float scale,
                             const float* rope_cos, const float* rope_sin, const int32_t* rope_pos,
                             const int32_t* seg_lo, const int32_t* seg_hi, float drop_p, uint64_t drop_seed, const float* rp_gate,
                             const float* rp_tab, int64_t rp_T, int64_t rp_ld, float* rp_ds, float* d_gate, float* d_tab, void* stream) {
  SLAM_CHECK_ARG(Q && K && V && O && dO && LSE && Delta && dQ && dK && dV, "slam_attn_bwd: null pointer");
  {
    const int need = slam_attn_needs_transposed(B, Tq, Tk, Hq, Hkv, D, ldq, ldk, ldv, lddo, (drop_p > 0.f ? 1 : 0) | (rp_gate ? 2 : 0));
    SLAM_CHECK_ARG(!(need & 2) || (Qt && Kt && dOt), "slam_attn_bwd: this configuration runs the kernels that read the transposed copies: "
                   "Qt / Kt / dOt must be given (slam_attn_needs_transposed says when)");
  }
  SLAM_CHECK_ARG(!rp_gate || (rp_tab && rp_ds && d_gate && d_tab && D == 64 && !causal && !seg_lo && !rope_cos && Hq == Hkv &&
                              rp_T >= Tq && rp_T >= Tk && rp_ld >= 2 * rp_T - 1),
                 "slam_attn_bwd: the gated relative position bias needs rp_tab / rp_ds / d_gate / d_tab, head_dim 64, bidirectional unpacked MHA "
                 "without fused RoPE, and a table covering the sequence");
  SLAM_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "slam_attn_bwd: drop_p=%f must be in [0, 1)", (double)drop_p);
  SLAM_CHECK_ARG(drop_p == 0.f || (D == 64 && !causal && !seg_lo && !rope_cos),
                 "slam_attn_bwd: attention-probability dropout is implemented for head_dim 64, bidirectional, unpacked batches");
  SLAM_CHECK_ARG((rope_cos == nullptr) == (rope_sin == nullptr), "slam_attn_bwd: rope_cos/rope_sin must both be set or both null");
  SLAM_CHECK_ARG(!rope_cos || Tq == Tk, "slam_attn_bwd: the fused RoPE gradient needs self-attention (Tq == Tk)");
  SLAM_CHECK_ARG((seg_lo == nullptr) == (seg_hi == nullptr), "slam_attn_bwd: seg_lo/seg_hi must both be set or both null");
  SLAM_CHECK_ARG(!seg_lo || (causal && Tq == Tk), "slam_attn_bwd: packed sequences need causal self-attention");
  if (int rc = check_common("slam_attn_bwd", B, Tq, Tk, Tqp, Tkp, Hq, Hkv, D, causal)) return rc;
  SLAM_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0 && ldo % 2 == 0 &&
                     lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0,
                 "slam_attn_bwd: leading dims misaligned");
  AttnParams p = {};
  p.Q = (const bf16_t*)Q; p.ldq = ldq; p.K = (const bf16_t*)K; p.ldk = ldk; p.V = (const bf16_t*)V; p.ldv = ldv;
  p.Qt = (const bf16_t*)Qt; p.Kt = (const bf16_t*)Kt; p.dOt = (const bf16_t*)dOt;
  p.O = (bf16_t*)O; p.ldo = ldo; p.dO = (const bf16_t*)dO; p.lddo = lddo;
  p.dQ = (bf16_t*)dQ; p.lddq = lddq; p.dK = (bf16_t*)dK; p.lddk = lddk; p.dV = (bf16_t*)dV; p.lddv = lddv;
  p.LSE = (float*)LSE; p.Delta = Delta; p.kmask = key_mask;
  p.Tq = (int)Tq; p.Tk = (int)Tk; p.Tqp = (int)Tqp; p.Tkp = (int)Tkp; p.Hq = (int)Hq; p.Hkv = (int)Hkv; p.scale = scale;
  p.rope_cos = rope_cos; p.rope_sin = rope_sin; p.rope_pos = rope_pos; p.seg_lo = seg_lo; p.seg_hi = seg_hi;
  p.drop_thresh = drop_p > 0.f ? slam_drop_thresh16(drop_p) : 0u;
  p.drop_scale = 1.0f / (1.0f - drop_p);
  p.drop_seed = drop_seed;
  p.drop_salt = g_slam_drop_salt;
  p.rp_gate = rp_gate; p.rp_tab = rp_tab; p.rp_T = (int)rp_T; p.rp_ld = (int)rp_ld; p.rp_ds = rp_ds;
  hipStream_t s = (hipStream_t)stream;
  const AttnFits fits = attn_fits(B, Tq, Tk, Tqp, Tkp, Hq, Hkv, D, ldq, ldk, ldv, lddo);
  if (rp_gate) {   // WavLM (unfrozen): the bias joins the recomputed scores; dL/d(score) is materialised once and reduced twice
    dim3 gq_((unsigned)cdiv64(Tq, 64), (unsigned)Hq, (unsigned)B), gk_((unsigned)cdiv64(Tk, 64), (unsigned)Hkv, (unsigned)B);
    if (drop_p > 0.f) {     // train-mode attention_dropout: the same mask as the forward, recomputed; dL/d(score) already carries it
      attn_launch((attn_bwd_dq_kernel<64, false, true, 1, true>), gq_, 256, 0, s, p);
      attn_launch((attn_bwd_dkdv_kernel<64, false, true, true>), gk_, 256, 0, s, p);
    } else {
      attn_launch((attn_bwd_dq_kernel<64, false, false, 1, true>), gq_, 256, 0, s, p);
      attn_launch((attn_bwd_dkdv_kernel<64, false, false, true>), gk_, 256, 0, s, p);
    }
    const int64_t rows = B * Hq * Tq;
    hipLaunchKernelGGL(relpos_dgate_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(rows, 4), 65535 * 4)), dim3(256), 0, s, rp_ds, rp_tab, d_gate, (int)B,
                       (int)Hq, (int)Tq, (int)Tk, (int)Tkp, (int)Tqp, (int)rp_T, (int)rp_ld);
    hipLaunchKernelGGL(relpos_dtab_kernel, dim3((unsigned)cdiv64(2 * rp_T - 1, 256), (unsigned)Hq), dim3(256), 0, s, rp_ds, rp_gate, d_tab, (int)B,
                       (int)Hq, (int)Tq, (int)Tk, (int)Tkp, (int)Tqp, (int)rp_T, (int)rp_ld);
    SLAM_CHECK_LAUNCH("slam_attn_bwd");
    return 0;
  }
  if (drop_p > 0.f) {   // same mask as the forward, recomputed (round-1 dK/dV kernel: the ring kernel has no dropout form)
    dim3 gq_((unsigned)cdiv64(Tq, 64), (unsigned)Hq, (unsigned)B), gk_((unsigned)cdiv64(Tk, 64), (unsigned)Hkv, (unsigned)B);
    attn_launch((attn_bwd_dq_kernel<64, false, true>), gq_, 256, 0, s, p);
    attn_launch((attn_bwd_dkdv_kernel<64, false, true>), gk_, 256, 0, s, p);
    SLAM_CHECK_LAUNCH("slam_attn_bwd");
    return 0;
  }
  dim3 gk2((unsigned)cdiv64(Tk, 128), (unsigned)Hkv, (unsigned)B);
  int rc = 0;
  if (g_attn_bwd_variant > 10 && D == 128 && causal) {   // timing ablations of the D = 128 causal ring kernel (tools only)
    SLAM_CHECK_ARG(fits.rowmajor && fits.copies, "slam_attn_bwd: the timing-ablation variants need tensors within the descriptors' 2 GiB");
    if ((rc = launch_dq<128, true>(p, B, s, fits))) return rc;
    switch (g_attn_bwd_variant) {
      case 11: rc = launch_dkdv_ring<128, true, 1>(p, gk2, s); break;
      case 12: rc = launch_dkdv_ring<128, true, 2>(p, gk2, s); break;
      case 19: rc = launch_dkdv_ring<128, true, 9>(p, gk2, s); break;
      case 14: rc = launch_dkdv_ring<128, true, 0>(p, gk2, s); break;
      default: rc = launch_dkdv_ring<128, true, 5>(p, gk2, s); break;
    }
    if (rc) return rc;
    SLAM_CHECK_LAUNCH("slam_attn_bwd");
    return 0;
  }
  if (D == 64) {
    if (causal) {
      if ((rc = launch_dq<64, true>(p, B, s, fits))) return rc;
      rc = launch_dkdv<64, true>(p, B, s, fits);
    } else {
      if ((rc = launch_dq<64, false>(p, B, s, fits))) return rc;
      rc = launch_dkdv<64, false>(p, B, s, fits);
    }
  } else {
    if (causal) {
      if ((rc = launch_dq<128, true>(p, B, s, fits))) return rc;
      rc = launch_dkdv<128, true>(p, B, s, fits);
    } else {
      if ((rc = launch_dq<128, false>(p, B, s, fits))) return rc;
      rc = launch_dkdv<128, false>(p, B, s, fits);
    }
  }
  if (rc) return rc;
  SLAM_CHECK_LAUNCH("slam_attn_bwd");
  return 0;
}
