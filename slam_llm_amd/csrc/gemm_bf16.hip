// bf16 "NT" GEMM on MFMA for gfx950:  C[M,N] = epilogue(alpha * A[M,K] . B[N,K]^T)
//
// Every dense contraction on the SLAM hot path is routed through this one kernel
// (reference sites: openai-whisper Linear/Conv1d called from
// src/slam_llm/models/encoder.py:18-29, projector Linear src/slam_llm/models/projector.py:24-26,
// HF LlamaDecoderLayer linears + lm_head via src/slam_llm/models/slam_model.py:400).
// Frozen weights are kept in HBM in BOTH orientations (W and W^T; 288 GB makes that free), so the
// backward dX = dY.W is again an NT product against W^T and no NN/TN variants exist.
//
// Structure (MI355X-first, see /opt/skills/guides/cdna_hip_programming.md section 5):
//   * BMxBN output tile per workgroup, BK = 64, waves arranged WM x WN, each wave owns a
//     (BM/WM)x(BN/WN) sub-tile of 16x16 MFMA fragments (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//   * A/B K-tiles are DMA'd HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), two LDS
//     stages, one barrier per K-tile; the next tile's DMA is in flight during the MFMA phase.
//   * LDS rows are 128 B (64 bf16); the 16-byte chunk index is XOR-swizzled with (row & 7).  The DMA
//     destination must stay lane-linear, so the swizzle is applied to the per-lane *source* address
//     and again on the ds_read_b128 side (same involution) -> conflict-free fragment reads.
//   * operands are swapped in the MFMA (B-fragment as the "A" operand) so each lane ends up with
//     4 consecutive output columns -> 8-byte bf16 / 16-byte fp32 stores, float4 bias loads.
//   * workgroup ids are remapped XCD-aware (block b runs on XCD b%8): each XCD gets a contiguous
//     run of tiles, ordered in groups of 8 M-tiles, so A/B panels are re-used out of that XCD's L2.
#include "common.h"
#include "gemm_common.h"
#include <atomic>
#include <type_traits>

namespace {

// tools only: shader-clock / real-time stamps of workgroup 0 (effective clock of a launch = d(shader cycles) / d(100 MHz ticks))
__device__ unsigned long long g_clk_probe[6];   // workgroup 0: {cycles, 100 MHz ticks} at entry and at k-loop end; cycles at k-loop start and after the epilogue

// split-K tail workspace (registered by the host with slam_gemm_set_workspace: the library never allocates): [SK_CNT_BYTES of
// arrival counters, zero | fp32 slabs].  One workspace per process: GEMM launches that may split must be ordered on one stream
// (they are: every product of the step is launched on torch's current stream).
constexpr int64_t SK_CNT_BYTES = 4096;
constexpr int SK_MAX_TILES = (int)(SK_CNT_BYTES / 4);
std::atomic<void*> g_gemm_ws{nullptr};
std::atomic<int64_t> g_gemm_ws_bytes{0};
// -1 off (DEFAULT: on MI355X the hand-off -- slab stores, L2 write-back, ticket, fix-up reads, ~35 us -- costs as much as the idle
// part of the last round saves on every product of the C3 / C2 / C4 steps, profiles/r03_gemm_splitk.md), 0 auto plan, >= 2 forced
// number of K slices (tools / tests)
std::atomic<int> g_gemm_splitk{-1};
// Round 4: K-sliced launch + reduce launch for MID-M products (fewer 256 x 256 tiles than half the CUs: M = 672 of C4, the small-batch
// recipes): 1 = the auto rule may pick it (DEFAULT), 0 = never, 2 = forced plans (302..316) take the two-launch form as well (sweeps)
std::atomic<int> g_gemm_sk2{1};
std::atomic<int> g_gemm_splitk_rmax{32};   // auto plan: split only when the last round holds <= rmax tiles ...
std::atomic<int> g_gemm_splitk_smax{2};    // ... into at most smax slices (slam_gemm_set_config 320 + rmax / 8, 340 + smax: sweeps)

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_nt_kernel(GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int FM = WTM / 16, FN = WTN / 16;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int NIA = BM / 8 / NW;  // DMA instructions per wave per K-tile for A
  constexpr int NIB = BN / 8 / NW;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/wave mismatch");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- XCD-aware tile remap (bijective for any grid size) ----
  const int nwg = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  bid = gemm_order_pos(p, nwg, bid & 7, bid >> 3);
  int tm, tn;
  gemm_tile_of(p, bid, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-lane DMA source pointers (row clamp handles the M/N edges) ----
  const int srow = lane >> 3;                // row within the 8-row DMA group
  const int schunk = (lane & 7) ^ srow;      // swizzled source chunk
  const bf16_t* a_src[NIA];
  const bf16_t* b_src[NIB];
#pragma unroll
  for (int j = 0; j < NIA; j++) {
    int r = m0 + (j * NW + wave) * 8 + srow;
    r = min(r, p.M - 1);
    a_src[j] = p.A + (int64_t)r * p.lda + schunk * 8;
  }
#pragma unroll
  for (int j = 0; j < NIB; j++) {
    int r = n0 + (j * NW + wave) * 8 + srow;
    r = min(r, p.N - 1);
    b_src[j] = p.B + (int64_t)r * p.ldb + schunk * 8;
  }

  auto stage = [&](int kt, int s) {
    char* sa = smem + s * STAGE;
    char* sb = sa + BM * ROWB;
    const int koff = kt * BK;
#pragma unroll
    for (int j = 0; j < NIA; j++) glds16(a_src[j] + koff, sa + (j * NW + wave) * 1024);
#pragma unroll
    for (int j = 0; j < NIB; j++) glds16(b_src[j] + koff, sb + (j * NW + wave) * 1024);
  };

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; i++)
#pragma unroll
    for (int j = 0; j < FN; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = base + lane&15, chunk = ks*4 + lane>>4, swizzled by row&7
  const int frow = lane & 15;
  const int fg = lane >> 4;

  const int nt = p.K / BK;
  stage(0, 0);
  for (int t = 0; t < nt; t++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nt) stage(t + 1, (t + 1) & 1);
    const char* sa = smem + (t & 1) * STAGE;
    const char* sb = sa + BM * ROWB;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ks++) {
      bf16x8_t af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; i++) {
        const int row = wm * WTM + i * 16 + frow;
        const int ch = (ks * 4 + fg) ^ (row & 7);
        af[i] = *reinterpret_cast<const bf16x8_t*>(sa + row * ROWB + ch * 16);
      }
#pragma unroll
      for (int j = 0; j < FN; j++) {
        const int row = wn * WTN + j * 16 + frow;
        const int ch = (ks * 4 + fg) ^ (row & 7);
        bfr[j] = *reinterpret_cast<const bf16x8_t*>(sb + row * ROWB + ch * 16);
      }
#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int j = 0; j < FN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  gemm_epilogue<FM, FN, WTM, WTN>(p, acc, m0, n0, wm, wn, frow, fg);
}


// ------------------------------------------------------------------------------------------------------------
// Pipelined variant: same tile/LDS geometry, but the K-tile is computed in two 32-deep phases whose operand
// fragments are double-buffered in REGISTERS.  That frees the LDS stage one phase earlier, so with only two LDS
// stages the DMA runs two K-tiles ahead, and the DMA issue / ds_reads are interleaved between the MFMAs
// (sched_group_barrier) instead of being serialised in front of them:
//   phase A(t): 32 MFMA on ks=0 fragments (regs)  ||  ds_read ks=1 fragments of tile t
//   vmcnt(0) ; barrier                            (stage t&1 is now free, tile t+1 is visible)
//   phase B(t): 32 MFMA on ks=1 fragments         ||  DMA tile t+2 -> stage t&1  ||  ds_read ks=0 fragments of tile t+1
// ------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int VAR, bool PROBE = false>   // PROBE (tools): workgroup 0 stamps g_clk_probe
__global__ __launch_bounds__(WM* WN * 64) void gemm_nt_pipe_kernel(GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int FM = WTM / 16, FN = WTN / 16;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int NIA = BM / 8 / NW;
  constexpr int NIB = BN / 8 / NW;
  static_assert(BK == 64, "two 32-deep phases per K-tile");
  // VAR: 0/1 schedule A/B reference (1 = shipped); 20/30/40/50 = timing ablations used by tools/gemm_bench.py
  // (no DMA / no fragment reads / neither / DMA never waited for: wrong results by design)
  constexpr bool NODMA = (VAR == 20 || VAR == 40), NOFRAG = (VAR == 30 || VAR == 40);

  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (PROBE && blockIdx.x == 0 && threadIdx.x == 0) {
    g_clk_probe[0] = __builtin_readcyclecounter();
    g_clk_probe[1] = wall_clock64();
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int nwg = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  bid = gemm_order_pos(p, nwg, bid & 7, bid >> 3);
  int tm, tn;
  gemm_tile_of(p, bid, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ srow;
  const bf16_t* a_src[NIA];
  const bf16_t* b_src[NIB];
#pragma unroll
  for (int j = 0; j < NIA; j++) {
    int r = min(m0 + (j * NW + wave) * 8 + srow, p.M - 1);
    a_src[j] = p.A + (int64_t)r * p.lda + schunk * 8;
  }
#pragma unroll
  for (int j = 0; j < NIB; j++) {
    int r = min(n0 + (j * NW + wave) * 8 + srow, p.N - 1);
    b_src[j] = p.B + (int64_t)r * p.ldb + schunk * 8;
  }
  auto stage = [&](int kt, int s) {
    char* sa = smem + s * STAGE;
    char* sb = sa + BM * ROWB;
    const int koff = kt * BK;
#pragma unroll
    for (int j = 0; j < NIA; j++) glds16(a_src[j] + koff, sa + (j * NW + wave) * 1024);
#pragma unroll
    for (int j = 0; j < NIB; j++) glds16(b_src[j] + koff, sb + (j * NW + wave) * 1024);
  };

  const int frow = lane & 15;
  const int fg = lane >> 4;
  // per-lane LDS byte offsets of the fragments (row part); the chunk part depends on ks only
  int a_off[FM], b_off[FN], a_sw[FM], b_sw[FN];
#pragma unroll
  for (int i = 0; i < FM; i++) {
    const int row = wm * WTM + i * 16 + frow;
    a_off[i] = row * ROWB;
    a_sw[i] = row & 7;
  }
#pragma unroll
  for (int j = 0; j < FN; j++) {
    const int row = wn * WTN + j * 16 + frow;
    b_off[j] = BM * ROWB + row * ROWB;
    b_sw[j] = row & 7;
  }
  auto load_frags = [&](const char* st, int ks, bf16x8_t (&af)[FM], bf16x8_t (&bfr)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; i++)
      af[i] = *reinterpret_cast<const bf16x8_t*>(st + a_off[i] + (((ks * 4 + fg) ^ a_sw[i]) << 4));
#pragma unroll
    for (int j = 0; j < FN; j++)
      bfr[j] = *reinterpret_cast<const bf16x8_t*>(st + b_off[j] + (((ks * 4 + fg) ^ b_sw[j]) << 4));
  };

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; i++)
#pragma unroll
    for (int j = 0; j < FN; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  bf16x8_t a0[FM], b0[FN], a1[FM], b1[FN];
  const int nt = p.K / BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (nt > 1) stage(1, 1);
  load_frags(smem, 0, a0, b0);
  if constexpr (NOFRAG) load_frags(smem, 1, a1, b1);

  if (PROBE && blockIdx.x == 0 && threadIdx.x == 0) g_clk_probe[4] = __builtin_readcyclecounter();
  for (int t = 0; t < nt; t++) {
    const int cur = t & 1;
    const char* st = smem + cur * STAGE;
    // ---- phase A: ks = 0 MFMAs, prefetch ks = 1 fragments of this tile ----
    if constexpr (!NOFRAG) load_frags(st, 1, a1, b1);   // VAR >= 20: timing ablations (wrong results by design, tools only)
#pragma unroll
    for (int i = 0; i < FM; i++)
#pragma unroll
      for (int j = 0; j < FN; j++)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j], a0[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < FM + FN; g++) {  // one ds_read per two MFMAs, remaining MFMAs trail
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
    if constexpr (VAR >= 1) __builtin_amdgcn_sched_barrier(0);  // keep phase A's MFMAs in front of the barrier
    if constexpr (VAR == 50) {   // ablation: DMA issued but never waited for (reads race with it: wrong results)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // ---- phase B: ks = 1 MFMAs, DMA of tile t+2 into the stage just released, ks = 0 fragments of tile t+1 ----
    // branch-free tail: past the last tile the DMA re-fetches tile nt-1 into the (already released) stage and the
    // fragment prefetch reads data nobody consumes -- keeps the whole iteration one schedulable basic block
    if constexpr (!NODMA) stage(min(t + 2, nt - 1), cur);
    if constexpr (!NOFRAG) load_frags(smem + (cur ^ 1) * STAGE, 0, a0, b0);
#pragma unroll
    for (int i = 0; i < FM; i++)
#pragma unroll
      for (int j = 0; j < FN; j++)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a1[i], acc[i][j], 0, 0, 0);
    // measured on MI355X: DMA issue first, fragment prefetch second beats the reverse order by 3-5 %, and s_setprio
    // around the MFMA runs is a loss on this schedule (profiles/r01_gemm_pmc.md)
#pragma unroll
    for (int g = 0; g < NIA + NIB; g++) {  // DMA issue spread over the first MFMAs
      __builtin_amdgcn_sched_group_barrier(0x010, 1, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
    }
#pragma unroll
    for (int g = 0; g < FM + FN; g++) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
    }
  }
  if (PROBE && blockIdx.x == 0 && threadIdx.x == 0) {
    g_clk_probe[2] = __builtin_readcyclecounter();
    g_clk_probe[3] = wall_clock64();
  }
  gemm_epilogue<FM, FN, WTM, WTN>(p, acc, m0, n0, wm, wn, frow, fg);
  if (PROBE && blockIdx.x == 0 && threadIdx.x == 0) g_clk_probe[5] = __builtin_readcyclecounter();
}

// ------------------------------------------------------------------------------------------------------------
// Persistent variant of the pipelined kernel (auto rule: K <= 2048): one workgroup per CU walks a strided list of output
// tiles and treats (tile, k-tile) as ONE stream -- the DMA of the next tile's first two k-tiles is issued in the last two
// phase-B slots of the current tile and its first fragments are read right after the epilogue, so the pipeline never
// drains: no exposed prologue (2 dependent HBM/L2 round trips per tile) and the epilogue's stores retire under the next
// tile's MFMAs.  The LDS-DMA goes through buffer descriptors (buffer_load_dwordx4 ... offen lds): the per-lane part of every source
// address is ONE 32-bit byte offset per operand computed before the loop, the tile / k-tile / row-group part is a scalar
// added to it (1 VALU per 1 KiB DMA instead of the ~4 of a clamped 64-bit pointer), and rows past M / N are handled by the
// descriptor's bounds check (they read as zeros) instead of per-lane clamps.  The byte offset is carried in the VGPR
// offset, which the hardware range-checks against num_records (the SGPR offset is not part of that check on gfx9).
// Requires (M + BM) * lda * 2 < 2^32 and (N + BN) * ldb * 2 < 2^32 (the launcher falls back to the one-tile kernel otherwise).
// ------------------------------------------------------------------------------------------------------------

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_nt_persist2_kernel(GemmParams p, unsigned bytes_a, unsigned bytes_b) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int FM = WTM / 16, FN = WTN / 16;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int NIA = BM / 8 / NW;
  constexpr int NIB = BN / 8 / NW;
  static_assert(BK == 64, "two 32-deep phases per K-tile");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int nwg = p.tiles_m * p.tiles_n;
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ srow;

  const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, bytes_a, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, bytes_b, 0x00020000);
  const unsigned lda2 = (unsigned)p.lda * 2u, ldb2 = (unsigned)p.ldb * 2u;   // row pitch in bytes
  // per-lane byte offset of this lane's 16-byte chunk inside an 8-row DMA group (the swizzle lives in the SOURCE chunk)
  const unsigned voff_a = (unsigned)(wave * 8 + srow) * lda2 + (unsigned)schunk * 16u;
  const unsigned voff_b = (unsigned)(wave * 8 + srow) * ldb2 + (unsigned)schunk * 16u;

  auto tile_origin = [&](int vbid, int& m0, int& n0) {  // same XCD-aware bijection as the one-tile kernels
    const int bid = gemm_order_pos(p, nwg, vbid & 7, vbid >> 3);
    int tm, tn;
    gemm_tile_of(p, bid, tm, tn);
    m0 = tm * BM;
    n0 = tn * BN;
  };
  auto stage = [&](int mo, int no, int kt, int s) {
    char* sa = smem + s * STAGE;
    char* sb = sa + BM * ROWB;
    // scalar parts, pinned to SGPRs (readfirstlane also stops the compiler from re-associating them into a per-lane multiply)
    const unsigned sa0 = __builtin_amdgcn_readfirstlane((unsigned)mo * lda2 + (unsigned)kt * (BK * 2));
    const unsigned sb0 = __builtin_amdgcn_readfirstlane((unsigned)no * ldb2 + (unsigned)kt * (BK * 2));
#pragma unroll
    for (int j = 0; j < NIA; j++)
      blds16(srd_a, voff_a + __builtin_amdgcn_readfirstlane(sa0 + (unsigned)(j * NW * 8) * lda2), sa + (j * NW + wave) * 1024);
#pragma unroll
    for (int j = 0; j < NIB; j++)
      blds16(srd_b, voff_b + __builtin_amdgcn_readfirstlane(sb0 + (unsigned)(j * NW * 8) * ldb2), sb + (j * NW + wave) * 1024);
  };

  const int frow = lane & 15;
  const int fg = lane >> 4;
  static_assert(WTM % 8 == 0 && WTN % 8 == 0, "wave tile bases must keep row & 7 == frow & 7");
  const int a_base = (wm * WTM + frow) * ROWB;
  const int b_base = BM * ROWB + (wn * WTN + frow) * ROWB;
  const int sw = frow & 7;
  auto load_frags = [&](const char* st, int ks, bf16x8_t (&af)[FM], bf16x8_t (&bfr)[FN]) {
    const int ch = ((ks * 4 + fg) ^ sw) << 4;
#pragma unroll
    for (int i = 0; i < FM; i++) af[i] = *reinterpret_cast<const bf16x8_t*>(st + a_base + ch + i * 16 * ROWB);
#pragma unroll
    for (int j = 0; j < FN; j++) bfr[j] = *reinterpret_cast<const bf16x8_t*>(st + b_base + ch + j * 16 * ROWB);
  };

  f32x4_t acc[FM][FN];
  bf16x8_t a0[FM], b0[FN], a1[FM], b1[FN];
  const int nt = p.K / BK;   // >= 2 (launcher)
  int vbid = blockIdx.x;
  int m0, n0;
  tile_origin(vbid, m0, n0);
  stage(m0, n0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  stage(m0, n0, 1, 1);
  load_frags(smem, 0, a0, b0);
  int par = 0;  // LDS stage holding the stream's current k-tile

  while (true) {
    const int vnext = vbid + gridDim.x;
    const bool has_next = vnext < nwg;
    int m1 = m0, n1 = n0;
    if (has_next) tile_origin(vnext, m1, n1);   // == current tile when there is none: the tail re-fetches harmless data
#pragma unroll
    for (int i = 0; i < FM; i++)
#pragma unroll
      for (int j = 0; j < FN; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto ktile = [&](int t, auto last_tag) {
      constexpr bool LAST = decltype(last_tag)::value;
      const int cur = par;
      const char* st = smem + cur * STAGE;
      load_frags(st, 1, a1, b1);
#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int j = 0; j < FN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j], a0[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < FM + FN; g++) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const bool nx = t + 2 >= nt;
      stage(nx ? m1 : m0, nx ? n1 : n0, nx ? t + 2 - nt : t + 2, cur);
      if constexpr (!LAST) load_frags(smem + (cur ^ 1) * STAGE, 0, a0, b0);
#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int j = 0; j < FN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a1[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NIA + NIB; g++) {
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
      }
      if constexpr (!LAST) {
#pragma unroll
        for (int g = 0; g < FM + FN; g++) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        }
      }
      par ^= 1;
    };
    for (int t = 0; t < nt - 1; t++) ktile(t, std::false_type{});
    ktile(nt - 1, std::true_type{});
    gemm_epilogue<FM, FN, WTM, WTN>(p, acc, m0, n0, wm, wn, frow, fg);
    if (!has_next) break;
    vbid = vnext;
    m0 = m1;
    n0 = n1;
    load_frags(smem + par * STAGE, 0, a0, b0);   // next tile's k-tile 0 (landed before the last barrier)
  }
}

// ------------------------------------------------------------------------------------------------------------
// 4-wave variant with a HAND-ORDERED k-loop (cfg 12).  256x256x64 tiles like the pipelined kernel, but 2 x 2 waves of 128x128
// (one wave per SIMD, 64 accumulators = 256 AGPRs): a wave tile twice as large halves the LDS bytes read per MFMA (128 KB
// instead of 192 KB per workgroup and K-tile).  With one wave per SIMD nothing hides a stall, so the instruction ORDER is the
// kernel: every MFMA, ds_read, LDS-DMA and wait of the loop is an `asm volatile` statement (the compiler only allocates
// registers; it cannot re-serialise the stream, which is what sank the compiler-scheduled attempt of round 1, -30 %):
//   phase A(t): 64 MFMA on the ks=0 fragments, one ds_read_b128 of a ks=1 fragment after every 4th
//   s_waitcnt vmcnt(0); s_barrier           tile t+1 has landed, stage t&1 is free
//   phase B(t): 64 MFMA on the ks=1 fragments, one LDS-DMA piece of tile t+2 after every 4th, one ds_read of a ks=0 fragment of
//               tile t+1 after every 4th (offset by two)
// Two K-tiles per loop trip so that the LDS stage is a compile-time constant of each half.
// ------------------------------------------------------------------------------------------------------------
// REG = false: operand tiles by LDS-DMA (one instruction per KiB, but 60-180 issue cycles each, which a lone wave per SIMD
// cannot hide).  REG = true: global -> registers (buffer_load_dwordx4, issued in phase A of tile t for tile t+2) -> ds_write_b128
// in phase B, after the barrier that releases the stage: two cheap instructions per KiB and 64 staging VGPRs.
// SK: the instantiation that carries the split-K tail (launched only when a split plan is active: the plain instantiation stays
// byte-for-byte the kernel the headline number is quoted on -- the tail code costs it registers and a few spilled loop invariants)
// SK: 0 = whole tiles only, 1 = in-launch split-K tail (ticket + last arriver; round 3), 2 = two-launch form (round 4): the slices only
// write their slabs -- no fence, no ticket, no fix-up code in this instantiation -- and gemm_sk_reduce_kernel adds them up
template <int BM, int BN, bool REG, int ABL = 0, bool PROBE = false, int SK = 0>   // ABL (tools, wrong results): 1 no operand traffic in the loop, 2 no fragment reads, 3 neither; PROBE (tools): workgroup 0 stamps g_clk_probe
__global__ __launch_bounds__(256) void gemm_nt_w4_kernel(GemmParams p, unsigned bytes_a, unsigned bytes_b) {
  if (PROBE && blockIdx.x == 0 && threadIdx.x == 0) {
    g_clk_probe[0] = __builtin_readcyclecounter();
    g_clk_probe[1] = wall_clock64();
  }
  constexpr int WTM = BM / 2, WTN = BN / 2;          // 128 x 128 per wave
  constexpr int FM = WTM / 16, FN = WTN / 16;        // 8 x 8 fragments
  constexpr int STAGE = (BM + BN) * ROWB;            // 64 KiB
  constexpr int NIA = BM / 8 / 4, NIB = BN / 8 / 4;  // 1 KiB DMA pieces per wave and K-tile: 8 + 8
  static_assert(BK == 64 && FM == 8 && FN == 8 && NIA == 8 && NIB == 8, "schedule below is written for 256x256x64, 4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- workgroup -> tile (XCD-aware bijection: XCD x owns the contiguous run [base(x), base(x + 1)) of the tile order) ----
  // Split-K tail ("stream-K" for the one-tile-per-workgroup launch): every 256x256 tile costs the same, so T tiles on 256 CUs are
  // ceil(T / 256) rounds and the last round holds only R = T mod 256 tiles (16 of 2576 for the dL/dh product of the C3 step: 240
  // CUs idle for a whole tile time).  With sk_S >= 2 the launch carries sk_main = T - R whole-tile workgroups first (the first
  // sk_main / 8 entries of every XCD's run) and then R * S workgroups that each compute 1 / S of the k-tiles of one of the
  // remaining R tiles -- they are dispatched last, fill the idle CUs of the last round, write fp32 partial slabs, take a ticket,
  // and the LAST arriver of a tile sums the S slabs in index order (bit-reproducible, whoever it is) and runs the epilogue.
  // No workgroup ever waits for another one: nothing here depends on co-residency or dispatch order.
  const int nwg = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  int sk_tile = -1, sk_split = 0;   // tail bookkeeping (wave-uniform)
  {
    const int q = nwg >> 3, r = nwg & 7;
    int xcd = bid & 7, idx = bid >> 3;
    if (SK && p.sk_S > 1 && bid >= p.sk_main) {
      const int w = bid - p.sk_main;
      sk_tile = w % p.sk_R;             // consecutive workgroups: different tiles, the same K slice
      sk_split = w / p.sk_R;
      xcd = sk_tile & 7;                // tail tile t is entry sk_main / 8 + t / 8 of XCD (t % 8)'s run
      idx = (p.sk_main >> 3) + (sk_tile >> 3);
    }
    bid = gemm_order_pos(p, nwg, xcd, idx);
  }
  int tm, tn;
  gemm_tile_of(p, bid, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  if (SK && sk_tile >= 0 && p.M - m0 <= 16) {   // a thin tile among the tail tiles is not split: slice 0 computes all of it below
    if (sk_split > 0) return;
    sk_tile = -1;
  }

  // ---- thin M-tail tile (at most 16 valid rows: 31 x 380 = 11780 tokens leave FOUR rows for the 47th row of tiles, 2.1 % of all
  // tiles of every LLM product).  B has no reuse over 16 rows, so nothing is staged: each wave takes 64 of the 256 columns and
  // feeds the MFMAs straight from global memory, 8 k-steps in flight; no LDS, no barrier.  ~a quarter of a full tile's time. ----
  if (p.M - m0 <= 16) {
    const int frow = lane & 15, fg = lane >> 4;
    const bf16_t* ap = p.A + (int64_t)min(m0 + frow, p.M - 1) * p.lda + fg * 8;
    const bf16_t* bp[4];
    // act 4 (fused SwiGLU forward, interleaved [gate64 | up64] column blocks): a wave takes 32 gate columns (fragments 0, 1) and
    // the 32 matching up columns (fragments 2, 3) of block wave >> 1 instead of 64 consecutive columns
    const int thin_c0 = (p.act == 4) ? n0 + (wave >> 1) * 128 + (wave & 1) * 32 : n0 + wave * 64;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int col = (p.act == 4) ? thin_c0 + (j & 1) * 16 + (j >> 1) * 64 : thin_c0 + j * 16;
      bp[j] = p.B + (int64_t)min(col + frow, p.N - 1) * p.ldb + fg * 8;
    }
    f32x4_t acc[1][4];
#pragma unroll
    for (int j = 0; j < 4; j++) acc[0][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int UN = 8;   // k-steps of 32 per trip (K % 64 == 0; the remainder loop takes the odd 64s)
    int k = 0;
    for (; k + 32 * UN <= p.K; k += 32 * UN) {
      bf16x8_t af[UN], bf[UN][4];
#pragma unroll
      for (int u = 0; u < UN; u++) {
        af[u] = *reinterpret_cast<const bf16x8_t*>(ap + k + 32 * u);
#pragma unroll
        for (int j = 0; j < 4; j++) bf[u][j] = *reinterpret_cast<const bf16x8_t*>(bp[j] + k + 32 * u);
      }
#pragma unroll
      for (int u = 0; u < UN; u++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[u][j], af[u], acc[0][j], 0, 0, 0);
    }
    for (; k < p.K; k += 32) {
      const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(ap + k);
#pragma unroll
      for (int j = 0; j < 4; j++)
        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(bp[j] + k), af, acc[0][j], 0, 0, 0);
    }
    if (p.act == 4) {
      // lane holds rows m0 + frow, columns thin_c0 + j * 16 + fg * 4 + 0..3 (gate, j = 0, 1) and + 64 (up); 8-byte stores
      const int m = m0 + frow;
      if (m < p.M && thin_c0 < p.N) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int n = thin_c0 + j * 16 + fg * 4;
          uint2 og, ou, oh;
          og.x = pack2bf(acc[0][j][0], acc[0][j][1]); og.y = pack2bf(acc[0][j][2], acc[0][j][3]);
          ou.x = pack2bf(acc[0][j + 2][0], acc[0][j + 2][1]); ou.y = pack2bf(acc[0][j + 2][2], acc[0][j + 2][3]);
          float hv[4];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const unsigned gw = (e >> 1) ? og.y : og.x, uw = (e >> 1) ? ou.y : ou.x;
            const float gf = __uint_as_float((e & 1) ? (gw & 0xFFFF0000u) : (gw << 16));
            const float uf = __uint_as_float((e & 1) ? (uw & 0xFFFF0000u) : (uw << 16));
            hv[e] = gf * sigmoid_fast(gf) * uf;
          }
          oh.x = pack2bf(hv[0], hv[1]); oh.y = pack2bf(hv[2], hv[3]);
          bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + n;
          *reinterpret_cast<uint2*>(c) = og;
          *reinterpret_cast<uint2*>(c + 64) = ou;
          // h column of interleaved column n (gate half of block n / 128): 64 * (n / 128) + n % 128
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C2) + (int64_t)m * p.ldc2 + ((n >> 7) << 6) + (n & 127)) = oh;
        }
      }
      return;
    }
    gemm_epilogue_generic<1, 4, 16, 64>(p, acc, m0, n0, 0, wave, frow, fg);
    return;
  }

  // DMA: piece j of this wave covers tile rows (j*4 + wave)*8 .. +7; lane -> (row = lane>>3, 16-byte chunk (lane&7) ^ row)
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ srow;
  const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, bytes_a, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, bytes_b, 0x00020000);
  // this workgroup's k-tiles: all of them, or slice sk_split of sk_S (the slice start is folded into the per-lane offsets)
  const int nt_all = p.K / BK;
  const int kt0 = (SK && sk_tile >= 0) ? (int)((int64_t)sk_split * nt_all / p.sk_S) : 0;
  const int kt1 = (SK && sk_tile >= 0) ? (int)((int64_t)(sk_split + 1) * nt_all / p.sk_S) : nt_all;
  unsigned a_vo[NIA], b_vo[NIB];
#pragma unroll
  for (int j = 0; j < NIA; j++) a_vo[j] = (unsigned)(((int64_t)min(m0 + (j * 4 + wave) * 8 + srow, p.M - 1) * p.lda + schunk * 8 + (int64_t)kt0 * BK) * 2);
#pragma unroll
  for (int j = 0; j < NIB; j++) b_vo[j] = (unsigned)(((int64_t)min(n0 + (j * 4 + wave) * 8 + srow, p.N - 1) * p.ldb + schunk * 8 + (int64_t)kt0 * BK) * 2);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
  const unsigned dma_a = lds0 + (unsigned)(wave * 1024), dma_b = dma_a + (unsigned)(BM * ROWB);

  // fragment reads: row (lane & 15) of fragment i of this wave's panel, chunk (ks*4 + lane>>4) ^ (row & 7)
  const int frow = lane & 15, fg = lane >> 4;
  unsigned fa[2][2], fb[2][2];   // per-lane byte address [stage][ks] (the ds_read immediate only reaches 64 KiB: one stage)
#pragma unroll
  for (int st = 0; st < 2; st++)
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const unsigned ch = (unsigned)(((ks * 4 + fg) ^ (frow & 7)) << 4);
      fa[st][ks] = lds0 + (unsigned)(st * STAGE + (wm * WTM + frow) * ROWB) + ch;
      fb[st][ks] = lds0 + (unsigned)(st * STAGE + BM * ROWB + (wn * WTN + frow) * ROWB) + ch;
    }

  // four 4x4 quadrants: hipcc leaves ONE 8x8 array of f32x4 (1 KiB) in scratch
  f32x4_t acc00[4][4], acc01[4][4], acc10[4][4], acc11[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc00[i][j] = acc01[i][j] = acc10[i][j] = acc11[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto mfma_ij = [&](auto i, auto j, const bf16x8_t& bfrag, const bf16x8_t& afrag) {
    if constexpr (i < 4 && j < 4) w4_mfma(acc00[i][j], bfrag, afrag);
    else if constexpr (i < 4) w4_mfma(acc01[i][j - 4], bfrag, afrag);
    else if constexpr (j < 4) w4_mfma(acc10[i - 4][j], bfrag, afrag);
    else w4_mfma(acc11[i - 4][j - 4], bfrag, afrag);
  };
  bf16x8_t a0[FM], b0[FN], a1[FM], b1[FN];
  const int nt = kt1 - kt0;

  auto dma_tile = [&](int kt, int stage) {   // whole tile, back to back (prologue only)
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(kt * BK * 2));
    gemm_static_for<0, NIA>([&](auto j) { w4_dma(srd_a, a_vo[j], so, __builtin_amdgcn_readfirstlane(dma_a + (unsigned)(stage * STAGE + j * 4096))); });
    gemm_static_for<0, NIB>([&](auto j) { w4_dma(srd_b, b_vo[j], so, __builtin_amdgcn_readfirstlane(dma_b + (unsigned)(stage * STAGE + j * 4096))); });
  };
  bf16x8_t stg[NIA + NIB];   // REG: one K-tile share of this wave in flight from global memory
  const unsigned wr0 = lds0 + (unsigned)(wave * 1024 + lane * 16);   // REG: lane-linear image of a 1 KiB piece, like the DMA's
  // (issuing k-tiles 0 AND 1 before the first wait -- both stages are free at the start, `s_waitcnt vmcnt(16)` -- was measured:
  // +0.5 % on four C3 shapes, -3 % on 11780 x 4096 x 4096, the step 392.4 / 393.2 vs 390.2 / 393.2 ms: neutral, not kept;
  // profiles/r03_gemm_prologue_ab.jsonl)
  dma_tile(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (nt > 1) dma_tile(1, 1);
  gemm_static_for<0, FN>([&](auto j) { w4_lds_read<j * 16 * ROWB>(b0[j], fb[0][0]); });
  gemm_static_for<0, FM>([&](auto i) { w4_lds_read<i * 16 * ROWB>(a0[i], fa[0][0]); });
  if constexpr (REG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the loop's counted vmcnt waits assume only ITS loads are in flight)

  // one K-tile; CUR = its LDS stage (compile time).  MFMA order: i outer, j inner -> a fragment is needed every 8th MFMA, all
  // b fragments by the first eight: reads are issued b first.
  auto tile = [&](int t, auto cur_c) {
    constexpr int CUR = decltype(cur_c)::value;
    constexpr unsigned SO = CUR * STAGE;
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(min(t + 2, nt - 1) * BK * 2));
    // ---- phase A ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    gemm_static_for<0, FM>([&](auto i) {
      gemm_static_for<0, FN>([&](auto j) {
        mfma_ij(i, j, b0[j], a0[i]);
        constexpr int n = i * FN + j;          // after MFMA n: every 4th slot a ks=1 fragment read (b first, then a)
        if constexpr (n % 2 == 0 && n < 32 && !(ABL & 2)) {  // the 16 ks=1 fragment reads ride on the FIRST half: all landed by the phase's end
          constexpr int r = n / 2;             // 0..15
          if constexpr (r < FN) w4_lds_read<r * 16 * ROWB>(b1[r], fb[CUR][1]);
          else w4_lds_read<(r - FN) * 16 * ROWB>(a1[r - FN], fa[CUR][1]);
        }
        if constexpr (REG && n % 4 == 3 && !(ABL & 1)) {     // tile t+2 -> staging registers (past the end: tile nt-1 again)
          constexpr int r = n / 4;
          if constexpr (r < NIA) w4_gload(stg[r], srd_a, a_vo[r], so);
          else w4_gload(stg[r], srd_b, b_vo[r - NIA], so);
        }
      });
    });
    // DMA form: tile t+1 has landed.  Both forms: this wave's ks=1 fragment reads are done, and (REG) its ds_writes of tile t+1
    // -- issued in the previous phase B -- are in LDS
    if constexpr (!REG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- phase B ----
    gemm_static_for<0, FM>([&](auto i) {
      gemm_static_for<0, FN>([&](auto j) {
        mfma_ij(i, j, b1[j], a1[i]);
        constexpr int n = i * FN + j;
        if constexpr (n % 4 == 1 && !(ABL & 1)) {            // tile t+2 into the stage just released (past the end: tile nt-1 again)
          constexpr int r = n / 4;
          if constexpr (REG) {                 // piece r left global memory one phase ago: wait for it alone, write it
            w4_vmwait<NIA + NIB - 1 - r>();
            if constexpr (r < NIA) w4_lds_write<r * 4096>(wr0 + SO, stg[r]);
            else w4_lds_write<BM * ROWB + (r - NIA) * 4096>(wr0 + SO, stg[r]);
          } else {
            if constexpr (r < NIA) w4_dma(srd_a, a_vo[r], so, __builtin_amdgcn_readfirstlane(dma_a + (unsigned)(SO + r * 4096)));
            else w4_dma(srd_b, b_vo[r - NIA], so, __builtin_amdgcn_readfirstlane(dma_b + (unsigned)(SO + (r - NIA) * 4096)));
          }
        }
        if constexpr (n % 2 == 0 && n < 32 && !(ABL & 2)) {  // ks=0 fragments of tile t+1 (landed: waited for before the barrier), first half
          constexpr int r = n / 2;
          if constexpr (r < FN) w4_lds_read<r * 16 * ROWB>(b0[r], fb[CUR ^ 1][0]);
          else w4_lds_read<(r - FN) * 16 * ROWB>(a0[r - FN], fa[CUR ^ 1][0]);
        }
      });
    });
  };
  if (PROBE && blockIdx.x == 0 && threadIdx.x == 0) g_clk_probe[4] = __builtin_readcyclecounter();
  // (a variant that stops fetching in the last two tiles -- three instantiations of the tile body -- made hipcc spill around the
  // asm statements: 876 bytes of scratch, wrong results (a spilled "=v" of a ds_read is stored before the data lands), removed)
  int t = 0;
  for (; t + 1 < nt; t += 2) {
    tile(t, std::integral_constant<int, 0>{});
    tile(t + 1, std::integral_constant<int, 1>{});
  }
  if (t < nt) tile(t, std::integral_constant<int, 0>{});
  // the compiler does not know the asm statements were MFMAs: cover the MFMA -> accumulator-read hazard and drain the
  // branch-free tail's DMA / reads (they target this workgroup's LDS) before the epilogue
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  if (PROBE && blockIdx.x == 0 && threadIdx.x == 0) {
    g_clk_probe[2] = __builtin_readcyclecounter();
    g_clk_probe[3] = wall_clock64();
  }
  if (SK && sk_tile >= 0) {
    // ---- split-K tail: publish this slice's partial sums, take a ticket; only the last arriver goes on to the epilogue ----
    // slab layout: float4 number ((quadrant * 16 + i * 4 + j) * 256 + tid): every wave store / load is one contiguous KiB
    float* slab = p.sk_ws + ((size_t)sk_tile * p.sk_S + sk_split) * (size_t)(BM * BN);
    auto put = [&](f32x4_t (&a)[4][4], int quad) {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          *reinterpret_cast<f32x4_t*>(slab + ((size_t)((quad * 16 + i * 4 + j) * 256 + tid) << 2)) = a[i][j];
    };
    put(acc00, 0); put(acc01, 1); put(acc10, 2); put(acc11, 3);
    if constexpr (SK == 2) return;   // two-launch form: the kernel boundary publishes the slabs, gemm_sk_reduce_kernel adds them up
    // producer side of the hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): every wave drains its stores, workgroup
    // barrier, ONE lane writes the XCD's L2 back (agent-scope release), asm vmcnt(0) (the compiler may drop the one the fence
    // implies), then the relaxed agent-scope ticket.  (Write-through `sc0 sc1` slab stores without the release were tried and
    // measured 2-4 % faster per launch, but the consumer's plain loads then read stale slabs in ~6 % of the elements -- removed;
    // profiles/r03_gemm_splitk.md.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(smem);   // (the k-loop is over: LDS is free)
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      *flag = __hip_atomic_fetch_add(p.sk_cnt + sk_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned ticket = *reinterpret_cast<volatile unsigned*>(flag);
    if (ticket != (unsigned)(p.sk_S - 1)) return;
    // consumer side: one agent-scope acquire (invalidates this CU's L1), barrier, plain loads
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      p.sk_cnt[sk_tile] = 0;   // ready for the next launch (published by the kernel boundary; nobody else touches it in this one)
    }
    __syncthreads();
    // fixed summation order s = 0 .. S-1 (this workgroup's own slab is re-read like the others: the result does not depend on
    // which slice happened to arrive last)
    const float* slab0 = p.sk_ws + (size_t)sk_tile * p.sk_S * (size_t)(BM * BN);
    // slab-major: the 64 loads a lane issues per slab are independent of each other (the first form walked the slabs per fragment
    // and paid one exposed round trip per slab and fragment: ~8 us per slab)
    auto get = [&](f32x4_t (&a)[4][4], int quad, const float* sp, bool first) {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const f32x4_t v = *reinterpret_cast<const f32x4_t*>(sp + ((size_t)((quad * 16 + i * 4 + j) * 256 + tid) << 2));
          a[i][j] = first ? v : a[i][j] + v;
        }
    };
    for (int sl = 0; sl < p.sk_S; sl++) {
      const float* sp = slab0 + (size_t)sl * (BM * BN);
      get(acc00, 0, sp, sl == 0); get(acc01, 1, sp, sl == 0); get(acc10, 2, sp, sl == 0); get(acc11, 3, sp, sl == 0);
    }
  }
  if (p.act == 4) {   // fused SwiGLU forward: the left / right quadrants of the wave tile are gate / up of the same 64 columns
    const int nb = n0 + wn * WTN, mb = m0 + wm * WTM;
    if (nb < p.N) {
      if (mb + WTM <= p.M) {
        gemm_epilogue_swiglu_fwd<true>(p, acc00, acc01, mb, nb, frow, fg);
        gemm_epilogue_swiglu_fwd<true>(p, acc10, acc11, mb + 64, nb, frow, fg);
      } else {
        gemm_epilogue_swiglu_fwd<false>(p, acc00, acc01, mb, nb, frow, fg);
        gemm_epilogue_swiglu_fwd<false>(p, acc10, acc11, mb + 64, nb, frow, fg);
      }
    }
  } else {
    gemm_epilogue<4, 4, WTM, WTN>(p, acc00, m0, n0, wm, wn, frow, fg);
    gemm_epilogue<4, 4, WTM, WTN>(p, acc01, m0, n0 + 64, wm, wn, frow, fg);
    gemm_epilogue<4, 4, WTM, WTN>(p, acc10, m0 + 64, n0, wm, wn, frow, fg);
    gemm_epilogue<4, 4, WTM, WTN>(p, acc11, m0 + 64, n0 + 64, wm, wn, frow, fg);
  }
  if (PROBE && blockIdx.x == 0 && threadIdx.x == 0) g_clk_probe[5] = __builtin_readcyclecounter();
}

// ------------------------------------------------------------------------------------------------------------
// Second launch of the two-launch split-K form (round 4): tail tile t = blockIdx.x / 4, quadrant blockIdx.x % 4 of the 4-wave kernel's
// 256 x 256 tile.  The slab layout is the kernel's own register layout (float4 number ((quadrant * 16 + i * 4 + j) * 256 + tid)), so
// thread tid adds up the SAME accumulators the kernel's thread tid held, slices in index order (bit-reproducible), and runs the
// kernel's epilogue on them.  Why two launches: with every tile of an under-filled grid sliced (C4's M = 672 products: 48 tiles on 256
// CUs) the in-launch form leaves the whole reduction to 48 last arrivers, ~9 us per 256 KiB slab each (672 x 4096 x 4096: 75 us
// unsplit, 98 us with five slices); here 4 x R workgroups read the slabs at the chip's streaming rate behind one kernel boundary.
// ------------------------------------------------------------------------------------------------------------
// (grid 16 x R since late round 4: a workgroup adds ONE 16-row fragment strip of a quadrant -- 4 float4 per thread and slice -- with the
// loads of up to eight slices in flight together; the 4 x R form walked the slices one dependent round trip at a time: 14 us per
// launch, 197 launches per C4 step)
__global__ __launch_bounds__(256) void gemm_sk_reduce_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256, WTM = 128, WTN = 128;
  const int t = blockIdx.x >> 4, quad = (blockIdx.x >> 2) & 3, strip = blockIdx.x & 3;
  const int nwg = p.tiles_m * p.tiles_n;
  int bid;
  {   // tail tile t -> position in the tile order: entry sk_main / 8 + t / 8 of XCD (t % 8)'s run (as in gemm_nt_w4_kernel)
    const int xcd = t & 7, idx = (p.sk_main >> 3) + (t >> 3);
    bid = gemm_order_pos(p, nwg, xcd, idx);
  }
  int tm_, tn_;
  gemm_tile_of(p, bid, tm_, tn_);
  const int m0 = tm_ * BM, n0 = tn_ * BN;
  if (p.M - m0 <= 16) return;   // thin tiles are never sliced: slice 0 computed and stored all of it
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, frow = lane & 15, fg = lane >> 4;
  const float* slab0 = p.sk_ws + (size_t)t * p.sk_S * (size_t)(BM * BN);
  f32x4_t acc[1][4];
  const float* sp0 = slab0 + ((size_t)((quad * 16 + strip * 4) * 256 + tid) << 2);
  for (int s0 = 0; s0 < p.sk_S; s0 += 8) {
    f32x4_t v[8][4];
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (s0 + u < p.sk_S) {
#pragma unroll
        for (int j = 0; j < 4; j++) v[u][j] = *reinterpret_cast<const f32x4_t*>(sp0 + (size_t)(s0 + u) * (BM * BN) + ((size_t)(j * 256) << 2));
      }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (s0 + u < p.sk_S) {       // slices in index order, as before: the same bits
#pragma unroll
        for (int j = 0; j < 4; j++) acc[0][j] = (s0 + u == 0) ? v[u][j] : acc[0][j] + v[u][j];
      }
  }
  gemm_epilogue<1, 4, WTM, WTN>(p, acc, m0 + (quad >> 1) * 64 + strip * 16, n0 + (quad & 1) * 64, wm, wn, frow, fg);
}

// ------------------------------------------------------------------------------------------------------------
// Tall-skinny products, N <= 64 (round 4): C[M, N] = A[M, K] . B[N, K]^T with K in the thousands -- the LoRA-extension columns of every
// dX product (du = dy . (s B)^T over K = 6144 ... 22016 output features) and the adapters' first hop where lora_a_fwd does not apply.
// One 128 x 64 tile column gave ceil(M / 128) workgroups walking ALL of K alone: 6 workgroups on 256 CUs at C4 (139 us, 8 TFLOP/s), 93 at
// C3 (80 us) -- for a product whose only cost is streaming A once (145 MB at C3: 23 us at the HBM rate).
// Here a workgroup is 64 rows x ONE K SLICE: 4 waves x 16 rows, operands straight from global memory into the MFMAs (A is read once, B
// is 64 rows that live in L1 / L2: nothing to stage), eight k-steps of loads in flight per wave; S slices per row block so that the grid
// is ~1000 workgroups whatever M is.  S = 1: the epilogue runs in place.  S > 1: fp32 partials [S][M][64] into the GEMM workspace, and
// gemm_ts_reduce_kernel adds them in slice order (bit-reproducible) and runs the generic epilogue.
// ------------------------------------------------------------------------------------------------------------
struct TsPlan { int S, nk; float* ws; };

__device__ __forceinline__ void ts_slice(int nk, int S, int sl, int& k0, int& k1) {
  k0 = (int)((int64_t)sl * nk / S) * 32;
  k1 = (int)((int64_t)(sl + 1) * nk / S) * 32;
}

__global__ __launch_bounds__(256) void gemm_ts_kernel(GemmParams p, TsPlan pl) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fg = lane >> 4;
  const int m0 = blockIdx.x * 64, sl = blockIdx.y;
  int k0, k1;
  ts_slice(pl.nk, pl.S, sl, k0, k1);
  const bf16_t* ap = p.A + (int64_t)min(m0 + wave * 16 + frow, p.M - 1) * p.lda + fg * 8;
  const bf16_t* bp[4];
#pragma unroll
  for (int j = 0; j < 4; j++) bp[j] = p.B + (int64_t)min(j * 16 + frow, p.N - 1) * p.ldb + fg * 8;
  f32x4_t acc[1][4];
#pragma unroll
  for (int j = 0; j < 4; j++) acc[0][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  constexpr int UN = 8;
  int k = k0;
  for (; k + 32 * UN <= k1; k += 32 * UN) {
    bf16x8_t af[UN], bf[UN][4];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      af[u] = *reinterpret_cast<const bf16x8_t*>(ap + k + 32 * u);
#pragma unroll
      for (int j = 0; j < 4; j++) bf[u][j] = *reinterpret_cast<const bf16x8_t*>(bp[j] + k + 32 * u);
    }
#pragma unroll
    for (int u = 0; u < UN; u++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[u][j], af[u], acc[0][j], 0, 0, 0);
  }
  for (; k < k1; k += 32) {
    const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(ap + k);
#pragma unroll
    for (int j = 0; j < 4; j++)
      acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(bp[j] + k), af, acc[0][j], 0, 0, 0);
  }
  if (pl.S == 1) {
    gemm_epilogue_generic<1, 4, 16, 64>(p, acc, m0, 0, wave, 0, frow, fg);
    return;
  }
  const int m = m0 + wave * 16 + frow;
  if (m >= p.M) return;
  float* wrow = pl.ws + ((size_t)sl * p.M + m) * 64;
#pragma unroll
  for (int j = 0; j < 4; j++) *reinterpret_cast<f32x4_t*>(wrow + j * 16 + fg * 4) = acc[0][j];
}

__global__ __launch_bounds__(256) void gemm_ts_reduce_kernel(GemmParams p, TsPlan pl) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fg = lane >> 4;
  const int m0 = blockIdx.x * (16 * (blockDim.x >> 6));   // launched with ONE wave per workgroup: 16 rows each, so that M / 16 CUs pull the slabs
  const int m = m0 + wave * 16 + frow;                    // (64-row workgroups: 11 CUs for the 672-row C4 products, each bound by its own fill rate)
  f32x4_t acc[1][4];
#pragma unroll
  for (int j = 0; j < 4; j++) acc[0][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (m < p.M) {
    // eight slices' loads in flight together (one dependent round trip per slice: 25 us for the 48 slices of the C4 shape, more than the
    // product itself); the sums stay in slice order
    for (int s0 = 0; s0 < pl.S; s0 += 8) {
      f32x4_t v[8][4];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (s0 + u < pl.S) {
          const float* wrow = pl.ws + ((size_t)(s0 + u) * p.M + m) * 64;
#pragma unroll
          for (int j = 0; j < 4; j++) v[u][j] = *reinterpret_cast<const f32x4_t*>(wrow + j * 16 + fg * 4);
        }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (s0 + u < pl.S) {
#pragma unroll
          for (int j = 0; j < 4; j++) acc[0][j] = (s0 + u == 0) ? v[u][j] : acc[0][j] + v[u][j];
        }
    }
  }
  gemm_epilogue_generic<1, 4, 16, 64>(p, acc, m0, 0, wave, 0, frow, fg);   // (all lanes: the 16-byte store path exchanges lane rows)
}

std::atomic<int> g_gemm_ts{1};   // 1 = N <= 64 products take the tall-skinny K-sliced kernel (DEFAULT), 0 = the 128 x 64 tile kernel (A/B)

// -> true when launched
static bool launch_gemm_ts(GemmParams& p, hipStream_t stream, int* rc) {
  // (large M: the staged 128 x 64 tile kernel streams A in full 128-byte lines and wins -- 56 vs 96 us at 11780 x 64 x 6144; this
  // kernel's fragment-shaped loads are 64-byte row segments.  It is the form for FEW rows and a long K: 102 -> 37 us at 672 x 64 x 12288)
  if (!g_gemm_ts || p.N > 64 || p.M > 4096 || p.act == 3 || p.act == 4 || p.K < 256) return false;
  const int rb = (p.M + 63) / 64;
  const int nk = p.K / 32;
  int S = 1024 / rb;
  if (S > nk / 8) S = nk / 8;        // >= 8 k-steps (256 of K: one batch of loads in flight) per slice
  if (S < 1) S = 1;
  float* ws = nullptr;
  if (S > 1) {
    char* base = (char*)g_gemm_ws.load();
    const int64_t need = SK_CNT_BYTES + (int64_t)S * p.M * 64 * 4;
    if (base == nullptr || need > g_gemm_ws_bytes) S = 1;   // no scratch registered: one slice per row block (still 64-row workgroups)
    else ws = reinterpret_cast<float*>(base + SK_CNT_BYTES);
  }
  TsPlan pl{S, nk, ws};
  hipLaunchKernelGGL(gemm_ts_kernel, dim3((unsigned)rb, (unsigned)S), dim3(256), 0, stream, p, pl);
  if (S > 1) hipLaunchKernelGGL(gemm_ts_reduce_kernel, dim3((unsigned)((p.M + 15) / 16)), dim3(64), 0, stream, p, pl);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    slam_set_error("slam_gemm_bf16_nt(tall-skinny): launch failed: %s", hipGetErrorString(e));
    *rc = -2;
  } else {
    *rc = 0;
  }
  return true;
}

template <int BM, int BN, bool REG, int ABL, bool PROBE, int SK>
int launch_gemm_w4_impl(GemmParams& p, int64_t nwg, hipStream_t stream) {
  constexpr int lds = 2 * (BM + BN) * ROWB;
  static std::atomic<bool> attr_set{false};   // (setting the attribute twice from two threads is harmless; the flag only saves the call)
  auto kern = gemm_nt_w4_kernel<BM, BN, REG, ABL, PROBE, SK>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      slam_set_error("gemm: cannot raise LDS limit to %d: %s", lds, hipGetErrorString(e));
      return -2;
    }
    attr_set = true;
  }
  const uint64_t bytes_a = ((uint64_t)(p.M - 1) * (uint64_t)p.lda + (uint64_t)p.K) * 2ull;
  const uint64_t bytes_b = ((uint64_t)(p.N - 1) * (uint64_t)p.ldb + (uint64_t)p.K) * 2ull;
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, stream, p, (unsigned)bytes_a, (unsigned)bytes_b);
  SLAM_CHECK_LAUNCH("slam_gemm_bf16_nt(4-wave, hand-ordered k-loop)");
  return 0;
}

template <int BM, int BN, bool REG, int ABL = 0, bool PROBE = false>
int launch_gemm_w4(GemmParams& p, hipStream_t stream, int want_two = 0) {   // want_two: slice count of the two-launch form picked by the auto rule (0 = none)
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  // ---- split-K tail plan (see the kernel): R = tiles of the last, partial round; S slices each so that R * S <= one round ----
  p.sk_main = 0; p.sk_R = 0; p.sk_S = 1; p.sk_ws = nullptr; p.sk_cnt = nullptr;
  bool two = false;
  int mode = g_gemm_splitk;   // -1 off, 0 auto, >= 2 forced slice count (tools / tests)
  if (want_two >= 2) mode = want_two;
  char* ws = (char*)g_gemm_ws.load();
  const int64_t ws_bytes = g_gemm_ws_bytes;
  if (!PROBE && ABL == 0 && (mode >= 0 || g_gemm_sk2) && ws != nullptr && nwg < (1ll << 30)) {
    const int n_cu = 256;
    const int nt = p.K / BK;
    const int R = (int)(nwg % n_cu);
    int S = 1;
    bool few_over = false;
    if (mode >= 2) S = mode;
    else if (mode < 0 && g_gemm_sk2 && R > 0 && R <= 8 && nwg > n_cu && nt >= 16 && p.act != 4) {
      // a handful of tiles over a whole number of rounds (C4's gate|up product: 3 x 86 = 258 tiles on 256 CUs -- a second round for
      // TWO tiles): those tiles run as 8 K slices each next to the last full round, two-launch form (167 -> 136 us)
      S = nt / 8 < 8 ? nt / 8 : 8;
      few_over = S >= 2;
    }
    else if (mode >= 0 && R > 0 && R <= g_gemm_splitk_rmax && nwg >= n_cu) {
      // auto: only a genuinely short last round (R <= rmax tiles), slices of >= 8 k-tiles, at most smax slabs to add up.  The
      // thresholds come from profiles/r03_gemm_splitk.md (tools/gemm_splitk_sweep.py on MI355X): the hand-off (slab stores,
      // L2 write-back, ticket, fix-up reads: ~35 us) has a fixed price that only a nearly empty last round pays back.
      S = n_cu / R;
      if (S > g_gemm_splitk_smax) S = g_gemm_splitk_smax;
      if (S > nt / 8) S = nt / 8;
    }
    const int Rt = (mode >= 2) ? (int)(nwg < n_cu ? nwg : (R ? R : n_cu)) : R;   // forced: also under-filled / exact grids
    const int64_t need = SK_CNT_BYTES + (int64_t)Rt * S * (int64_t)(BM * BN * 4);
    if (S >= 2 && Rt >= 1 && Rt <= SK_MAX_TILES && nt / S >= 1 && need <= ws_bytes && ((nwg - Rt) % 8) == 0) {
      p.sk_main = (int)(nwg - Rt); p.sk_R = Rt; p.sk_S = S;
      p.sk_cnt = reinterpret_cast<unsigned*>(ws);
      p.sk_ws = reinterpret_cast<float*>(ws + SK_CNT_BYTES);
      nwg = (int64_t)p.sk_main + (int64_t)Rt * S;
      two = (want_two >= 2 || few_over || (g_gemm_sk2 == 2 && mode >= 2)) && p.act != 4;
    }
  }
  if constexpr (!PROBE && ABL == 0) {
    if (p.sk_S > 1 && two) {
      const int rc = launch_gemm_w4_impl<BM, BN, REG, ABL, false, 2>(p, nwg, stream);
      if (rc == 0) {
        hipLaunchKernelGGL(gemm_sk_reduce_kernel, dim3((unsigned)(16 * p.sk_R)), dim3(256), 0, stream, p);
        SLAM_CHECK_LAUNCH("slam_gemm_bf16_nt(split-K reduce)");
      }
      return rc;
    }
    if (p.sk_S > 1) return launch_gemm_w4_impl<BM, BN, REG, ABL, false, 1>(p, nwg, stream);
  }
  return launch_gemm_w4_impl<BM, BN, REG, ABL, PROBE, 0>(p, nwg, stream);
}

// (A persistent form of this kernel -- one workgroup per CU, the branch-free tail fetching the NEXT output tile's first two
// k-tiles so that no prologue and no workgroup launch separates two tiles -- was built and measured: correct, 3-18 % SLOWER than
// the one-tile kernel on every shape (1223 vs 1386 TF at 11780 x 4096 x 4096, 949 vs 1156 at K = 1280): static striding loses the
// dispatcher's load balancing, and the epilogue inside the tile loop spills.  Removed; profiles/r02_gemm_experiments.md.)

// (Round 2 also built this loop on v_mfma_f32_32x32x16_bf16 -- 2 x 2 and 2 x 4 waves, operand tiles by LDS-DMA and by
// buffer_load -> ds_write with one and two k-tiles of register staging; all correct, all slower: 2740-2870 cycles per k-tile
// against an MFMA-only floor of 2065-2090, i.e. every VMEM instruction of the loop stalls the SIMD for 40-50 cycles whatever issues
// it, and the 16x16x32 stream interleaves with them better.  Numbers in profiles/r02_gemm_experiments.md; code in the history
// (commit "GEMM experiments: 32x32x16-MFMA hand-ordered kernels").)

template <int BM, int BN, int WM, int WN, int PIPE = -1, bool PROBE = false>
int launch_gemm(GemmParams& p, hipStream_t stream) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  constexpr int lds = 2 * (BM + BN) * ROWB;
  static std::atomic<bool> attr_set{false};
  void (*kern)(GemmParams);
  if constexpr (PIPE >= 0) kern = gemm_nt_pipe_kernel<BM, BN, WM, WN, PIPE, PROBE>;
  else kern = gemm_nt_kernel<BM, BN, WM, WN>;
  if (!attr_set && lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      slam_set_error("gemm: cannot raise LDS limit to %d: %s", lds, hipGetErrorString(e));
      return -2;
    }
    attr_set = true;
  }
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(WM * WN * 64), lds, stream, p);
  SLAM_CHECK_LAUNCH("slam_gemm_bf16_nt");
  return 0;
}

template <int BM, int BN, int WM, int WN>
int launch_gemm_persist2(GemmParams& p, hipStream_t stream) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  constexpr int lds = 2 * (BM + BN) * ROWB;
  static std::atomic<bool> attr_set{false};
  static std::atomic<int> n_cu{0};
  auto kern = gemm_nt_persist2_kernel<BM, BN, WM, WN>;
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
    n_cu = prop.multiProcessorCount;   // (before the flag: a thread that sees the flag sees the count)
    attr_set = true;
  }
  // exact extents of the two operand views: rows past the end are out of range for the descriptor and read as zeros
  const uint64_t bytes_a = ((uint64_t)(p.M - 1) * (uint64_t)p.lda + (uint64_t)p.K) * 2ull;
  const uint64_t bytes_b = ((uint64_t)(p.N - 1) * (uint64_t)p.ldb + (uint64_t)p.K) * 2ull;
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  const int64_t ncu = n_cu;
  const int64_t grid = nwg < ncu ? nwg : ncu;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WM * WN * 64), lds, stream, p, (unsigned)bytes_a, (unsigned)bytes_b);
  SLAM_CHECK_LAUNCH("slam_gemm_bf16_nt(persistent, descriptor DMA)");
  return 0;
}

// the descriptor form addresses every byte of a (row-padded) operand with 32 bits
// (extent of a row view whose rows may OVERLAP -- lda < K, the conv window views: (rows - 1) * ld + K elements -- plus one tile of rows
// past the end that the kernels address before the range check zeroes them)
static inline bool fits_descriptor(int64_t rows, int64_t ld, int tile, int64_t K) {
  return ((uint64_t)(rows + tile - 1) * (uint64_t)ld + (uint64_t)K) * 2ull < (1ull << 32);
}

// tuning state (tools / sweeps set it between launches; relaxed atomics: a launch reads each knob once, whole)
std::atomic<int> g_gemm_cfg{0};  // 0 = auto
std::atomic<int> g_gemm_big{12};       // which 256x256 kernel the auto rule uses for K > 2048 (6 | 7 | 12): in the C3 step 12 -> 401.7 ms, 6 -> 408.3 ms
std::atomic<int> g_gemm_big_shortk{7}; // ... and for K <= 2048
std::atomic<int> g_gemm_small{1};      // the 128x128 kernel of the auto rule: 1 = two-stage loop, 11 = register-double-buffered pipeline
std::atomic<int> g_gemm_group_m{0};    // 0 = per-shape rule (gemm_group_m_for), 1..64 = forced (tools)
std::atomic<int> g_gemm_probe{0};      // 1: cfg 6 / 12 launch their PROBE instantiation (workgroup 0 stamps g_clk_probe; tools/gemm_epi_probe.py)

}  // namespace

// Raster group height (M-tiles per XCD-local group; a pure renumbering of tiles: results are bit-identical whatever it is).  Round 6
// (tools/gemm_enc_raster.py on MI355X, profiles/r06_gemm_raster.md): 8 is at or within 0.5 % of the best for most products of the step, but three
// shape classes have a better value by 2-2.6 %: short-K products with many column tiles (Whisper / HuBERT qkv, fc1: 12), narrow outputs behind a
// long K (fc2: N = 1280, K = 5120: 4), and wide outputs with K <= 4096 (dX of down_proj, gate|up forward, lm_head: 4).
std::atomic<int> g_gemm_gm_rule[4] = {{12}, {4}, {4}, {8}};     // the rule's four values (slam_gemm_set_group_m_rule: sweeps)
static int gemm_group_m_for(int64_t M, int64_t N, int64_t K) {
  const int forced = g_gemm_group_m;
  if (forced > 0) return forced;
  const int64_t tn = (N + 255) / 256;
  if (K <= 2048 && tn >= 12) return g_gemm_gm_rule[0];
  if (K >= 4096 && tn <= 6) return g_gemm_gm_rule[1];
  if (K <= 4096 && tn >= 48) return g_gemm_gm_rule[2];
  return g_gemm_gm_rule[3];
}
extern "C" int slam_gemm_set_group_m_rule(int short_k_many_cols, int long_k_narrow, int wide_k4096, int other) {   // tools: the four values of the rule
  const int v[4] = {short_k_many_cols, long_k_narrow, wide_k4096, other};
  for (int i = 0; i < 4; i++) SLAM_CHECK_ARG(v[i] >= 1 && v[i] <= 64, "slam_gemm_set_group_m_rule: value %d out of range [1,64]", v[i]);
  for (int i = 0; i < 4; i++) g_gemm_gm_rule[i] = v[i];
  return 0;
}

extern "C" int slam_gemm_set_group_m(int group_m) {   // tuning knob (tools): raster group height; 0 = the per-shape rule (default)
  SLAM_CHECK_ARG(group_m >= 0 && group_m <= 64, "slam_gemm_set_group_m: %d out of range [0,64]", group_m);
  g_gemm_group_m = group_m;
  return 0;
}

extern "C" int slam_gemm_debug_clock(unsigned long long* out6) {   // tools: stamps of the last pipelined-kernel launch (sync first)
  SLAM_CHECK_ARG(out6 != nullptr, "slam_gemm_debug_clock: null output");
  hipError_t e = hipMemcpyFromSymbol(out6, HIP_SYMBOL(g_clk_probe), 6 * sizeof(unsigned long long));
  if (e != hipSuccess) {
    slam_set_error("slam_gemm_debug_clock: %s", hipGetErrorString(e));
    return -2;
  }
  return 0;
}

extern "C" int slam_gemm_set_workspace(void* workspace, int64_t bytes) {
  SLAM_CHECK_ARG((workspace == nullptr && bytes == 0) || (workspace != nullptr && bytes >= SK_CNT_BYTES + 2 * 256 * 256 * 4 && ((uintptr_t)workspace % 16) == 0),
                 "slam_gemm_set_workspace: need a 16-byte aligned device buffer of >= %ld bytes whose first %ld bytes are zero (or null, 0 to detach)",
                 (long)(SK_CNT_BYTES + 2 * 256 * 256 * 4), (long)SK_CNT_BYTES);
  g_gemm_ws_bytes = 0;
  g_gemm_ws = workspace;
  g_gemm_ws_bytes = bytes;
  return 0;
}

// every GEMM tuning knob back to the value it is DEFINED with above (slam_reset_tuning: one place for the defaults)
void slam_gemm_reset_tuning_() {
  g_gemm_splitk = -1; g_gemm_sk2 = 1; g_gemm_splitk_rmax = 32; g_gemm_splitk_smax = 2; g_gemm_ts = 1;
  g_gemm_cfg = 0; g_gemm_big = 12; g_gemm_big_shortk = 7; g_gemm_small = 1; g_gemm_group_m = 0; g_gemm_probe = 0;
  g_gemm_gm_rule[0] = 12; g_gemm_gm_rule[1] = 4; g_gemm_gm_rule[2] = 4; g_gemm_gm_rule[3] = 8;
}

extern "C" int slam_gemm_set_config(int cfg) {
  if (cfg >= 360 && cfg <= 362) { g_gemm_sk2 = cfg - 360; return 0; }       // two-launch split-K for mid-M products: off / auto (default) / also for forced plans
  if (cfg == 370 || cfg == 371) { g_gemm_ts = cfg - 370; return 0; }         // N <= 64 products: 128 x 64 tile kernel / tall-skinny K-sliced kernel (default)
  if (cfg >= 320 && cfg <= 336) { g_gemm_splitk_rmax = (cfg - 320) * 8; return 0; }   // auto plan: tail tiles <= 8 * (cfg - 320)
  if (cfg >= 340 && cfg <= 348) { g_gemm_splitk_smax = cfg - 340; return 0; }         // auto plan: at most cfg - 340 slices
  if (cfg >= 300 && cfg <= 316) {   // split-K tail of the 4-wave kernel: 300 = auto, 301 = off, 302..316 = forced slice count (tools / tests)
    g_gemm_splitk = (cfg == 300) ? 0 : (cfg == 301 ? -1 : cfg - 300);
    return 0;
  }
  // 100 + v / 200 + v (v in 6, 7, 12): the 256x256 kernel the AUTO rule picks for K > 2048 / K <= 2048 (tools/step sweeps)
  if (cfg == 400 || cfg == 401) { g_gemm_probe = cfg - 400; return 0; }   // tools: cycle stamps of workgroup 0 (slam_gemm_debug_clock)
  if (cfg == 106 || cfg == 107 || cfg == 112) { g_gemm_big = cfg - 100; return 0; }
  if (cfg == 206 || cfg == 207 || cfg == 212) { g_gemm_big_shortk = cfg - 200; return 0; }
  if (cfg == 601 || cfg == 611) { g_gemm_small = cfg - 600; return 0; }   // which 128x128 kernel the auto rule uses when small tiles win (1 | 11)
  SLAM_CHECK_ARG(cfg == 0 || cfg == 1 || cfg == 2 || cfg == 3 || cfg == 4 || cfg == 6 || cfg == 7 || cfg == 11 || cfg == 12, "slam_gemm_set_config: cfg %d (0 auto | 1 2 3 4 6 7 11 12)", cfg);
  g_gemm_cfg = cfg;
  return 0;
}

extern "C" int slam_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                                 int64_t ldc, int64_t M, int64_t N, int64_t K, const float* bias,
                                 const void* residual, int64_t ldr, int64_t res_row_mod, int act,
                                 float alpha, int out_dtype, int accumulate, void* stream) {
  SLAM_CHECK_ARG(A && B && C, "slam_gemm_bf16_nt: null operand");
  SLAM_CHECK_ARG(M > 0 && N > 0 && K > 0, "slam_gemm_bf16_nt: bad shape M=%ld N=%ld K=%ld", (long)M,
                 (long)N, (long)K);
  SLAM_CHECK_ARG(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "slam_gemm_bf16_nt: dim >= 2^31");
  SLAM_CHECK_ARG(K % 64 == 0, "slam_gemm_bf16_nt: K=%ld must be a multiple of 64 (pad the operands)", (long)K);
  SLAM_CHECK_ARG(N % 4 == 0, "slam_gemm_bf16_nt: N=%ld must be a multiple of 4", (long)N);
  SLAM_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0, "slam_gemm_bf16_nt: lda/ldb must be multiples of 8 elements");
  SLAM_CHECK_ARG(ldc % 4 == 0, "slam_gemm_bf16_nt: ldc must be a multiple of 4 elements");
  SLAM_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0,
                 "slam_gemm_bf16_nt: operands must be 16-byte aligned");
  // lda < K is allowed: the rows of A then OVERLAP -- a strided-convolution window over a row-major [T, C] signal is exactly such a
  // view (row t = the k * C contiguous elements from row stride * t on), so conv layers need no im2col copy; every kernel addresses
  // A as row * lda + k and sizes its descriptor as (M - 1) * lda + K elements
  SLAM_CHECK_ARG(lda >= 8 && ldb >= K && ldc >= N, "slam_gemm_bf16_nt: leading dimension too small");
  SLAM_CHECK_ARG(act >= 0 && act <= 3, "slam_gemm_bf16_nt: act %d unknown", act);
  if (act == 3) {
    SLAM_CHECK_ARG(residual && ldr >= 2 * N && ldc >= 2 * N && out_dtype == SLAM_BF16 && !accumulate && !bias,
                   "slam_gemm_bf16_nt: act 3 (swiglu backward) needs residual = [gate | up] with ldr >= 2N, bf16 C with ldc >= 2N, "
                   "no bias / accumulate");
  }
  SLAM_CHECK_ARG(out_dtype == SLAM_BF16 || out_dtype == SLAM_F32, "slam_gemm_bf16_nt: out_dtype %d unknown", out_dtype);
  if (residual) {
    SLAM_CHECK_ARG(ldr % 4 == 0 && ldr >= N && ((uintptr_t)residual % 8) == 0,
                   "slam_gemm_bf16_nt: bad residual layout");
  }
  GemmParams p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.bias = bias; p.res = (const bf16_t*)residual; p.ldr = ldr; p.res_mod = (int)res_row_mod;
  p.act = act; p.alpha = alpha; p.out_f32 = (out_dtype == SLAM_F32); p.accumulate = accumulate;
  p.C2 = nullptr; p.ldc2 = 0;
  int want_two = 0;
  p.group_m = gemm_group_m_for(M, N, K);
  hipStream_t s = (hipStream_t)stream;
  int cfg = g_gemm_cfg;
  if ((cfg == 0 || cfg == 3) && N <= 64) {
    int rc = 0;
    if (launch_gemm_ts(p, s, &rc)) return rc;
  }
  const int big = g_gemm_big, big_shortk = g_gemm_big_shortk;
  if (cfg == 0) {
    // auto (measured on MI355X, profiles/r01_perf_ops_first.json, tools/gemm_bench.py): the pipelined 256x256 tile
    // runs 1.25-1.4x the 128x128 one per CU; pick whichever loses less to wave quantisation over the 256 CUs
    // (256x256: 1 WG/CU -> 256 slots per round; 128x128: 2 WG/CU -> 512 slots).  E.g. the lm_head^T product of the
    // CE backward (4096 x 4096 x 128256) is exactly 256 big tiles = one full round.  Skinny N takes 128x64.
    const int64_t tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
    const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
    // time in units of "one CU doing one 128x128 tile": a 256x256 round = 4 tiles of work at ~1.3-1.4x speed (ties go to the big tile)
    const double t256 = (double)((tiles256 + 255) / 256) * (4.0 / 1.4);
    const double t128 = (double)((tiles128 + 511) / 512) * 2.0;
    // the persistent form (next tile's first two k-tiles DMA'd under the current tile's tail, no pipeline drain between
    // tiles) pays on short-K products, where prologue + epilogue are a visible share of a tile: +3-4 % at K = 1280
    // (Whisper), +-1 % at K >= 4096 (profiles/r02_gemm_experiments.md)
    if (N <= 64) cfg = 3;
    // 4-wave kernel: faster k-loop, longer epilogue (64 fragments per wave) -- it wins from K = 4096 up unless the output is narrow
    // (N = 1280 with a residual epilogue, Whisper fc2: 1084 vs 1124 TF for the 8-wave pipelined kernel, tools/gemm_enc_bench.py)
    else if (t256 < t128) cfg = (K <= 2048) ? big_shortk : ((big == 12 && N < 2048) ? 6 : big);
    else cfg = g_gemm_small;
    // mid-M (round 4): fewer 256 x 256 tiles than half the CUs and a K worth slicing -> the 4-wave kernel on K slices + the reduce
    // launch.  Cost model in the same unit (one CU x one 128 x 128 x K tile, ~22 us at K = 4096): 1 / S of a big tile's 2.86, plus the
    // slab stores, the launch boundary and the reduce pass (~20 us whatever K is) -- measured on the C4 shapes, tools/gemm_splitk_sweep.py
    if (g_gemm_sk2 && tiles256 <= 128 && N >= 256 && K >= 2048 && g_gemm_ws.load() != nullptr &&
        fits_descriptor(M, lda, 1, K) && fits_descriptor(N, ldb, 1, K)) {
      const int nt = (int)(K / BK);
      int S = (int)(256 / tiles256);
      if (S > nt / 8) S = nt / 8;
      if (S > 8) S = 8;
      const double tsk = (4.0 / 1.4) / S + 4500.0 / (double)K;
      if (S >= 2 && SK_CNT_BYTES + tiles256 * S * (int64_t)(256 * 256 * 4) <= g_gemm_ws_bytes && tsk < 0.95 * (t256 < t128 ? t256 : t128)) {
        cfg = 12;
        want_two = S;
      }
    }
  }
  switch (cfg) {
    case 1: return launch_gemm<128, 128, 2, 2>(p, s);
    case 11:                                               // 128x128 tiles on the register-double-buffered pipeline (two workgroups per CU)
      if (p.K < 2 * BK) return launch_gemm<128, 128, 2, 2>(p, s);
      return launch_gemm<128, 128, 2, 2, 1>(p, s);
    case 2: return launch_gemm<256, 128, 4, 2>(p, s);
    case 3: return launch_gemm<128, 64, 2, 2>(p, s);
    case 4: return launch_gemm<256, 256, 2, 4>(p, s);
    case 6:                                                // pipelined, phase A pinned before the barrier (shipped)
      if (g_gemm_probe) return launch_gemm<256, 256, 2, 4, 1, true>(p, s);
      return launch_gemm<256, 256, 2, 4, 1>(p, s);
    case 7:                                                // persistent pipelined, descriptor DMA (auto: short-K products)
      if (p.K < 2 * BK || !fits_descriptor(p.M, p.lda, 256, p.K) || !fits_descriptor(p.N, p.ldb, 256, p.K)) return launch_gemm<256, 256, 2, 4, 1>(p, s);
      return launch_gemm_persist2<256, 256, 2, 4>(p, s);
    // (round 5's cfg 8 -- 128 x 256 tiles, two workgroups per CU, hand-ordered, bit-identical to cfg 7 -- measured 26 % slower
    //  (profiles/r05_gemm_two_wg.md) and was removed in round 6 with its translation unit; git history keeps it)
    // (cfg 5 = compiler-placed barrier, 8-11 = timing ablations of the pipelined loop, 13 = register-staged 4-wave form,
    //  14-16 = ablations of the 4-wave loop: measured, recorded in profiles/r02_gemm_experiments.md, removed to keep the build short)
    case 12:                                               // 4 waves, hand-ordered k-loop
      if (p.K < 2 * BK || !fits_descriptor(p.M, p.lda, 1, p.K) || !fits_descriptor(p.N, p.ldb, 1, p.K))
        return launch_gemm<256, 256, 2, 4, 1>(p, s);
      if (g_gemm_probe) return launch_gemm_w4<256, 256, false, 0, true>(p, s);
      return launch_gemm_w4<256, 256, false>(p, s, want_two);
  }
  slam_set_error("slam_gemm_bf16_nt: bad config %d", cfg);
  return -1;
}

// ---- gate|up product with the SwiGLU forward in its epilogue (act 4) -----------------------------------------------------------
// Runs on the 4-wave kernel only (its wave tile is 128 columns = one [gate64 | up64] block).  slam_gemm_swiglu_supported() tells
// the host whether the AUTO rule would pick that kernel for the shape anyway and the operands fit its 32-bit descriptors; when it
// does not, the host runs the plain product + slam_swiglu_fwd.
extern "C" int slam_gemm_swiglu_supported(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb) {
  if (M <= 0 || N <= 0 || K < 2 * BK || K % 64 || N % 128 || K <= 2048 || N < 2048) return 0;
  if (g_gemm_cfg != 0 && g_gemm_cfg != 12) return 0;
  if (g_gemm_cfg == 0) {
    if (g_gemm_big != 12) return 0;
    const int64_t tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
    const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
    const double t256 = (double)((tiles256 + 255) / 256) * (4.0 / 1.4);
    const double t128 = (double)((tiles128 + 511) / 512) * 2.0;
    if (!(t256 < t128)) return 0;
  }
  if ((uint64_t)M * (uint64_t)lda * 2ull >= (1ull << 32) || (uint64_t)N * (uint64_t)ldb * 2ull >= (1ull << 32)) return 0;
  return 1;
}

extern "C" int slam_gemm_swiglu_bf16_nt(const void* A, int64_t lda, const void* B_interleaved, int64_t ldb, void* GU, int64_t ldgu,
                                        void* H, int64_t ldh, int64_t M, int64_t N, int64_t K, void* stream) {
  SLAM_CHECK_ARG(A && B_interleaved && GU && H, "slam_gemm_swiglu_bf16_nt: null operand");
  SLAM_CHECK_ARG(M > 0 && N > 0 && K > 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "slam_gemm_swiglu_bf16_nt: bad shape");
  SLAM_CHECK_ARG(K % 64 == 0 && N % 128 == 0, "slam_gemm_swiglu_bf16_nt: K=%ld must be a multiple of 64, N=%ld (= 2 F) of 128", (long)K, (long)N);
  SLAM_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldgu % 8 == 0 && ldh % 8 == 0 && lda >= K && ldb >= K && ldgu >= N && ldh >= N / 2,
                 "slam_gemm_swiglu_bf16_nt: leading dimensions must be multiples of 8 elements and cover the rows");
  SLAM_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B_interleaved % 16) == 0 && ((uintptr_t)GU % 16) == 0 && ((uintptr_t)H % 16) == 0,
                 "slam_gemm_swiglu_bf16_nt: operands must be 16-byte aligned");
  SLAM_CHECK_ARG(slam_gemm_swiglu_supported(M, N, K, lda, ldb) == 1,
                 "slam_gemm_swiglu_bf16_nt: shape %ld x %ld x %ld is not served by the 4-wave kernel (ask slam_gemm_swiglu_supported first)",
                 (long)M, (long)N, (long)K);
  GemmParams p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B_interleaved; p.C = GU;
  p.lda = lda; p.ldb = ldb; p.ldc = ldgu;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.bias = nullptr; p.res = nullptr; p.ldr = 0; p.res_mod = 0;
  p.act = 4; p.alpha = 1.0f; p.out_f32 = 0; p.accumulate = 0;
  p.C2 = H; p.ldc2 = ldh;
  p.group_m = gemm_group_m_for(M, N, K);
  return launch_gemm_w4<256, 256, false>(p, (hipStream_t)stream);
}
