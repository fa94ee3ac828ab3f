// bf16 "NT" GEMM, two workgroups per CU (cfg 8): see the kernel's header comment.  Launched by slam_gemm_bf16_nt (gemm_bf16.hip).
#include "common.h"
#include "gemm_common.h"
#include <atomic>

// ------------------------------------------------------------------------------------------------------------
// TWO WORKGROUPS PER CU (round 5, cfg 8; auto rule for the short-K Whisper encoder products).  The 256 x 256 persistent kernel above
// owns a CU alone: while its eight waves run a tile's epilogue (bias / GELU / residual, conversions, lane exchanges, 128 KiB of
// stores: 5-10 k cycles of a 52 k-cycle K = 1280 tile) and while they sit at the k-tile barrier or at a counted vmcnt, the matrix pipes
// of that CU idle -- 29 % of its wave cycles are parked and the pipes are busy 0.555 of the time at a clock that is NOT power limited
// (profiles/r04_pmc.md).  Here a workgroup is 4 waves (2 x 2 wave tiles of 64 x 128) on a 128 x 256 tile and TWO workgroups share a CU
// (two waves per SIMD, <= 256 registers each): one workgroup's epilogue, barrier and DMA waits are covered by the other's MFMAs.  The
// second-slot workgroups start half a tile late (s_sleep), so that the two epilogues of a CU never coincide.
//   * k-tiles of 32 (LDS rows of 64 bytes), 3-stage ring of (128 + 256) x 64 B = 24 KiB: 72 KiB per workgroup, 144 of the CU's 160 KiB;
//   * operand fragments double-buffered in registers: iteration t computes on the fragments of k-tile t while those of t + 1 are read,
//     so the stage of k-tile t is free at the barrier and the DMA runs THREE k-tiles ahead (two whole iterations of cover for a
//     counted vmcnt(6));  one barrier per k-tile;  (tile, k-tile) is one stream as in the kernel above: the first three k-tiles of
//     the next tile are requested in the last three iterations of the current one and the epilogue runs with them in flight;
//   * 16-byte chunk swizzle for 64-byte rows: row r keeps source chunk c in slot c ^ g((r >> 2) & 3), g = (0, 3, 2, 1): the four
//     16-lane groups of a ds_read_b128 (lanes {0-3, 12-15, 20-27}, ...) each touch 16 distinct 16-byte slots of the 256-byte bank
//     window (rows r, r + 12 with chunk fg and rows r + 4, r + 8 with chunk fg ^ 1 land in g(0), g(3), 1 ^ g(1), 1 ^ g(2) = 0 1 2 3);
//     applied on the DMA SOURCE address (the destination is lane-linear) and again on the read side;
//   * descriptor LDS-DMA as above: one 32-bit per-lane byte offset per operand, scalar tile / k parts, rows past M / N read as zeros.
// Costs against the 256 x 256 tile: 1.5x the L2 -> LDS bytes per FLOP ((128 + 256) / (128 x 256) vs 512 / 256^2), one barrier per 32
// MFMAs of a wave instead of per 64.  Requires K % 64 == 0 (an even number of k-tiles: the two fragment sets alternate), K >= 128.
// ------------------------------------------------------------------------------------------------------------
// Epilogue of the two-workgroup kernel.  The accumulators live in AGPRs (the MFMA statements say so); handed to gemm_epilogue as arrays
// the compiler copies ALL of them to VGPRs at the loop exit -- 128 values into the 128 arch VGPRs a two-waves-per-SIMD kernel has -- and
// spills them (500 bytes of scratch per lane and tile).  So one row of four fragments at a time is pulled out by explicit
// v_accvgpr_read statements and goes through the epilogue as a 16 x 64 "wave tile" (origin folded into m0 / n0, wm = wn = 0): 16 live
// values at a time.  The (bias, activation, residual) switch is taken ONCE, outside the row loop.
#define P3_ROWS(CALL)                                                                                                  \
  _Pragma("unroll") for (int h = 0; h < 2; h++) {                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                                    \
      f32x4_t row[1][4];                                                                                               \
      _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                                  \
        float e0, e1, e2, e3;                                                                                          \
        if (h == 0) {                                                                                                  \
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(e0) : "a"(accl[i][j][0]));                                   \
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(e1) : "a"(accl[i][j][1]));                                   \
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(e2) : "a"(accl[i][j][2]));                                   \
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(e3) : "a"(accl[i][j][3]));                                   \
        } else {                                                                                                       \
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(e0) : "a"(accr[i][j][0]));                                   \
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(e1) : "a"(accr[i][j][1]));                                   \
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(e2) : "a"(accr[i][j][2]));                                   \
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(e3) : "a"(accr[i][j][3]));                                   \
        }                                                                                                              \
        row[0][j] = f32x4_t{e0, e1, e2, e3};                                                                           \
      }                                                                                                                \
      const int mr = mw + i * 16, nc = nw + h * 64;                                                                    \
      CALL;                                                                                                            \
    }                                                                                                                  \
  }

__device__ __forceinline__ void p3_epilogue(const GemmParams& p, f32x4_t (&accl)[4][4], f32x4_t (&accr)[4][4], int mw, int nw, int frow, int fg) {
  if (p.out_f32 || p.accumulate || p.act == 3 || p.alpha != 1.0f) {
    P3_ROWS((gemm_epilogue_generic<1, 4, 16, 64>(p, row, mr, nc, 0, 0, frow, fg)))
    return;
  }
  const int key = (p.bias ? 1 : 0) | (p.res ? 2 : 0) | (p.act << 2);   // wave uniform
  const bool inner = (mw + 64 <= p.M) && (nw + 128 <= p.N);
#define P3_EPI(B_, A_, R_)                                                                                             \
  do {                                                                                                                 \
    if (inner) { P3_ROWS((gemm_epilogue_bf16<1, 4, 16, 64, B_, A_, R_, true>(p, row, mr, nc, 0, 0, frow, fg))) }       \
    else { P3_ROWS((gemm_epilogue_bf16<1, 4, 16, 64, B_, A_, R_, false>(p, row, mr, nc, 0, 0, frow, fg))) }            \
  } while (0)
  switch (key) {
    case 0: P3_EPI(false, 0, false); break;
    case 1: P3_EPI(true, 0, false); break;
    case 2: P3_EPI(false, 0, true); break;
    case 3: P3_EPI(true, 0, true); break;
    case 4: P3_EPI(false, 1, false); break;
    case 5: P3_EPI(true, 1, false); break;
    case 6: P3_EPI(false, 1, true); break;
    case 7: P3_EPI(true, 1, true); break;
    case 8: P3_EPI(false, 2, false); break;
    case 9: P3_EPI(true, 2, false); break;
    case 10: P3_EPI(false, 2, true); break;
    default: P3_EPI(true, 2, true); break;
  }
#undef P3_EPI
}
#undef P3_ROWS

constexpr int P3_BM = 128, P3_BN = 256, P3_KT = 32, P3_ROWB = 64, P3_NS = 3;
constexpr int P3_STAGE = (P3_BM + P3_BN) * P3_ROWB;   // 24 KiB

__global__ __launch_bounds__(256, 2) void gemm_nt_p3_kernel(GemmParams p, unsigned bytes_a, unsigned bytes_b, int stagger) {
  constexpr int FM = 4, FN = 8, WTM = 64, WTN = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = p.tiles_m * p.tiles_n;

  if (stagger > 0 && (int)blockIdx.x >= (int)(gridDim.x >> 1)) {   // second-slot workgroups: half a tile late (see above)
    for (int i = 0; i < stagger; i++) __builtin_amdgcn_s_sleep(127);
  }

  const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, bytes_a, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, bytes_b, 0x00020000);
  const unsigned lda2 = (unsigned)p.lda * 2u, ldb2 = (unsigned)p.ldb * 2u;
  // a 1 KiB DMA piece = 16 rows x 64 B, lane-linear in LDS: lane -> row lane >> 2, slot lane & 3, which holds SOURCE chunk slot ^ g
  const int drow = lane >> 2;
  const unsigned dchunk = (unsigned)((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3));
  const unsigned voff_a = (unsigned)(wave * 16 + drow) * lda2 + dchunk * 16u;
  const unsigned voff_b = (unsigned)(wave * 16 + drow) * ldb2 + dchunk * 16u;

  auto tile_origin = [&](int vbid, int& m0, int& n0) {  // the XCD-aware bijection of the other kernels
    const int xcd = vbid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int bid = base + (vbid >> 3);
    const int GM = p.group_m;
    const int per_group = GM * p.tiles_n;
    const int group = bid / per_group;
    const int first_m = group * GM;
    const int gsz = min(GM, p.tiles_m - first_m);
    const int within = bid - group * per_group;
    m0 = (first_m + within % gsz) * P3_BM;
    n0 = (within / gsz) * P3_BN;
  };
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
  // one k-tile's six pieces of this wave -> the stage at byte offset soff: A pieces w, w + 4 (rows (4 j + w) * 16 ...), B pieces w, w + 4,
  // w + 8, w + 12.  The row part of the source offset rides in the VGPR offset (range-checked: rows past M / N read as zeros), the
  // k part in the scalar offset.  PIECE = compile-time piece index 0..5 so that the loop can place the requests one by one.
  auto dma_piece = [&](auto piece_c, int mo, int no, unsigned koff, unsigned soff) {
    constexpr int PIECE = decltype(piece_c)::value;
    if constexpr (PIECE < 2) {
      const unsigned rows = __builtin_amdgcn_readfirstlane((unsigned)(mo + PIECE * 64) * lda2);
      w4_dma(srd_a, voff_a + rows, koff, __builtin_amdgcn_readfirstlane(lds0 + soff + (unsigned)((PIECE * 4 + wave) * 1024)));
    } else {
      constexpr int J = PIECE - 2;
      const unsigned rows = __builtin_amdgcn_readfirstlane((unsigned)(no + J * 64) * ldb2);
      w4_dma(srd_b, voff_b + rows, koff, __builtin_amdgcn_readfirstlane(lds0 + soff + (unsigned)(P3_BM * P3_ROWB + (J * 4 + wave) * 1024)));
    }
  };
  auto stage = [&](int mo, int no, int kt, unsigned soff) {   // a whole k-tile back to back (prologue only)
    const unsigned koff = __builtin_amdgcn_readfirstlane((unsigned)kt * (P3_KT * 2));
    gemm_static_for<0, 6>([&](auto pc) { dma_piece(pc, mo, no, koff, soff); });
  };

  const int frow = lane & 15;
  const int fg = lane >> 4;
  const unsigned fch = (unsigned)((fg ^ ((4 - ((frow >> 2) & 3)) & 3)) << 4);   // slot of source chunk fg in this lane's fragment row
  const unsigned a_base = lds0 + (unsigned)((wm * WTM + frow) * P3_ROWB) + fch;
  const unsigned b_base = lds0 + (unsigned)(P3_BM * P3_ROWB + (wn * WTN + frow) * P3_ROWB) + fch;

  // two 4 x 4 halves (columns 0..63 / 64..127 of the wave tile): one 4 x 8 array leaves the loop through scratch (the register allocator
  // spills all 128 accumulators at the loop exit), two halves are handed to the epilogue one after the other out of the AGPRs
  f32x4_t accl[FM][4], accr[FM][4];
  bf16x8_t a0[FM], b0[FN], a1[FM], b1[FN];
  const int nt = p.K / P3_KT;   // even, >= 4 (launcher)
  int vbid = blockIdx.x;
  int m0, n0;
  tile_origin(vbid, m0, n0);
  stage(m0, n0, 0, 0u);
  stage(m0, n0, 1, (unsigned)P3_STAGE);
  stage(m0, n0, 2, (unsigned)(2 * P3_STAGE));
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // k-tile 0 of this wave's pieces (6 per k-tile) has landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  gemm_static_for<0, FN>([&](auto j) { w4_lds_read<j * 16 * P3_ROWB>(b0[j], b_base); });
  gemm_static_for<0, FM>([&](auto i) { w4_lds_read<i * 16 * P3_ROWB>(a0[i], a_base); });
  unsigned soff = 0;   // byte offset of the stage holding the stream's current k-tile

  while (true) {
    const int vnext = vbid + gridDim.x;
    const bool has_next = vnext < nwg;
    int m1 = m0, n1 = n0;
    if (has_next) tile_origin(vnext, m1, n1);   // == the current tile when there is none: the tail re-fetches harmless data
#pragma unroll
    for (int i = 0; i < FM; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) accl[i][j] = accr[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // one k-tile: 32 MFMAs on (ax, bx); the six DMA requests of the k-tile three ahead behind MFMAs 0..5, the twelve fragment reads
    // of the next k-tile into (ay, by) behind MFMAs 7, 9, .. 29 (B fragments first: MFMA order is i outer, j inner)
    auto ktile = [&](int t, bf16x8_t (&ax)[FM], bf16x8_t (&bx)[FN], bf16x8_t (&ay)[FM], bf16x8_t (&by)[FN]) {
      // this wave's pieces of k-tile t + 1 have landed (t + 2's six may stay in flight) and its reads of k-tile t are complete ...
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();    // ... for every wave: k-tile t + 1 is visible, the stage of k-tile t is free
      asm volatile("" ::: "memory");
      const bool nx = t + 3 >= nt;
      const int mo = nx ? m1 : m0, no = nx ? n1 : n0;
      const unsigned koff = __builtin_amdgcn_readfirstlane((unsigned)(nx ? t + 3 - nt : t + 3) * (P3_KT * 2));
      const unsigned snext = (soff == (unsigned)(2 * P3_STAGE)) ? 0u : soff + (unsigned)P3_STAGE;
      const unsigned ra = a_base + snext, rb = b_base + snext;
      gemm_static_for<0, FM * FN>([&](auto n_c) {
        constexpr int n = decltype(n_c)::value;
        constexpr int i = n / FN, j = n % FN;
        if constexpr (j < 4) w4_mfma(accl[i][j], bx[j], ax[i]);
        else w4_mfma(accr[i][j - 4], bx[j], ax[i]);
        if constexpr (n < 6) {
          dma_piece(n_c, mo, no, koff, soff);
        } else if constexpr (n >= 7 && (n - 7) % 2 == 0 && (n - 7) / 2 < FM + FN) {
          constexpr int r = (n - 7) / 2;
          if constexpr (r < FN) w4_lds_read<r * 16 * P3_ROWB>(by[r], rb);
          else w4_lds_read<(r - FN) * 16 * P3_ROWB>(ay[r - FN], ra);
        }
      });
      soff = snext;
    };
    for (int t = 0; t < nt; t += 2) {
      ktile(t, a0, b0, a1, b1);
      ktile(t + 1, a1, b1, a0, b0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the next tile's first fragments are in a0 / b0; the epilogue may use the LDS counter)
    p3_epilogue(p, accl, accr, m0 + wm * WTM, n0 + wn * WTN, frow, fg);
    // A wait the COMPILER can see (the builtin, not an asm statement): its counter model carries the epilogue's guarded bias / residual
    // loads as "possibly pending" into the next tile's all-asm loop and would otherwise drop an `s_waitcnt vmcnt(0)` in front of the
    // first MFMA that reads a register one of those loads once targeted -- i.e. drain the DMA ring in the middle of every first
    // k-tile.  Here it costs the acknowledgement of this tile's last stores (the other workgroup of the CU computes meanwhile).
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
    if (!has_next) break;
    vbid = vnext;
    m0 = m1;
    n0 = n1;
    // the next tile's first fragments are read AGAIN here (the last iteration's prefetch of them is dropped): kept live across the
    // epilogue they cost 48 of the 128 arch VGPRs and the epilogue's accumulators went through scratch (500 bytes per lane)
    {
      const unsigned ra = a_base + soff, rb = b_base + soff;
      gemm_static_for<0, FN>([&](auto j) { w4_lds_read<j * 16 * P3_ROWB>(b0[j], rb); });
      gemm_static_for<0, FM>([&](auto i) { w4_lds_read<i * 16 * P3_ROWB>(a0[i], ra); });
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the tail's harmless re-fetches must not outlive the workgroup's LDS)
}

int launch_gemm_p3(GemmParams& p, hipStream_t stream, int stagger_mode) {
  p.tiles_m = (p.M + P3_BM - 1) / P3_BM;
  p.tiles_n = (p.N + P3_BN - 1) / P3_BN;
  constexpr int lds = P3_NS * P3_STAGE;
  static std::atomic<bool> attr_set{false};
  static std::atomic<int> n_cu{0};
  auto kern = gemm_nt_p3_kernel;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      slam_set_error("gemm: cannot raise LDS limit to %d: %s", lds, hipGetErrorString(e));
      return -2;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      slam_set_error("gemm: cannot query the device");
      return -2;
    }
    n_cu = prop.multiProcessorCount;
    attr_set = true;
  }
  const uint64_t bytes_a = ((uint64_t)(p.M - 1) * (uint64_t)p.lda + (uint64_t)p.K) * 2ull;
  const uint64_t bytes_b = ((uint64_t)(p.N - 1) * (uint64_t)p.ldb + (uint64_t)p.K) * 2ull;
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  const int64_t slots = 2 * (int64_t)n_cu;
  const int64_t grid = nwg < slots ? nwg : slots;
  p.group_m *= 2;   // 128-row tiles: the same rows of A per raster group as the 256-row kernels
  // half a tile in units of s_sleep 127 (~8 k cycles each): a k-tile of a workgroup that shares its SIMDs takes ~1.3 k cycles
  int stagger = 0;
  if (stagger_mode && grid > n_cu) stagger = (int)(((int64_t)(p.K / P3_KT) * 650 + 4000) / 8128);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, p, (unsigned)bytes_a, (unsigned)bytes_b, stagger);
  SLAM_CHECK_LAUNCH("slam_gemm_bf16_nt(two workgroups per CU)");
  return 0;
}

