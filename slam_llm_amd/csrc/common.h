// Shared device/host helpers for the slam_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SLAM_BF16 0
#define SLAM_F32 1

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---- error plumbing (host) -------------------------------------------------
extern "C" const char* slam_last_error();
void slam_set_error(const char* fmt, ...);
#define SLAM_CHECK_ARG(cond, ...)      \
  do {                                 \
    if (!(cond)) {                     \
      slam_set_error(__VA_ARGS__);     \
      return -1;                       \
    }                                  \
  } while (0)
#define SLAM_CHECK_LAUNCH(name)                                             \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      slam_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return -2;                                                            \
    }                                                                       \
  } while (0)

// ---- bf16 <-> f32 (device) -------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) {
  return __uint_as_float(((unsigned)v) << 16);
}
// round-to-nearest-even with the gfx950 hardware converter (v_cvt_pk_bf16_f32, NaN stays NaN): one instruction instead
// of the 6-op integer sequence -- the software form made RoPE / norm kernels VALU-heavy for no numerical difference
__device__ __forceinline__ bf16_t f2bf(float f) {
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
// two floats -> packed bf16x2 with the hardware converter (v_cvt_pk_bf16_f32 on gfx950, round-to-nearest-even)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  bf16x2_t v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}

// logistic function on the hardware's exp2 / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each, far below the bf16 rounding of every user):
// `1 / (1 + __expf(-x))` compiles to a range fix-up around the exponential and a division sequence, ~3x the instructions, which
// made the SwiGLU kernels VALU-bound just under the HBM rate.  x -> -inf gives rcp(inf) = 0, x -> +inf gives rcp(1) = 1.
__device__ __forceinline__ float sigmoid_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// exact-erf GELU.  erf via Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below the bf16 output rounding):
// one v_exp + one v_rcp + a 5-term Horner instead of libm's branchy erff in the epilogue of every fc1 tile.
// (v_exp_f32 directly: the argument is <= 0 and an underflow to 0 is the right answer, so __expf's range fix-up -- a compare, a
// select and two multiplies per element -- is dead weight; 128 elements per thread and tile go through this)
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float ex = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);   // exp(-z^2) = 2^(-x^2 / 2 * log2(e))
  const float e = fmaf(-poly * t, ex, 1.0f);                                  // erf(|x| / sqrt2)
  const float hx = 0.5f * x;
  return fmaf(hx, copysignf(e, x), hx);
}

// ---- wave / block reductions (wave = 64 lanes) -------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum for blockDim.x == NT (multiple of 64); red must hold NT/64 floats
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  if constexpr (NT == 64) return v;
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; i++) t += red[i];
  return t;
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  if constexpr (NT == 64) return v;
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; i++) t = fmaxf(t, red[i]);
  return t;
}

// ---- counter-based dropout mask (peft lora_dropout; shared by every kernel that applies or recomputes it) -------
// One splitmix64 hash serves the 4 consecutive elements of group (index >> 2), 16 random bits each:
//   keep(index) = bits16(index) >= round(p * 2^16).   `base` must be a multiple of 8; bit e of the result = keep(base+e).
__device__ __forceinline__ unsigned long long slam_mix64(unsigned long long z) {  // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// Dropout "salt" (round 6, for captured training steps): a captured hipGraph replays its kernel arguments, so a mask seed passed by value
// would repeat the SAME masks on every replay.  slam_set_dropout_salt(dev_ptr) registers one device-resident 64-bit word; every
// dropout-aware kernel XORs *dev_ptr into its seed when the pointer is non-null, and the captured step bumps the word at its end (forward
// and backward of one step read the same value).  null (default, eager mode): seeds are used as passed -- the masks the tests rebuild on
// the host.  Defined in capi_core.hip.
extern const unsigned long long* g_slam_drop_salt;
__device__ __forceinline__ unsigned long long slam_salted(unsigned long long seed, const unsigned long long* salt) {
  return salt ? (seed ^ *salt) : seed;
}

__device__ __forceinline__ unsigned slam_keep8(unsigned long long seed, unsigned long long base, unsigned thresh16) {
  unsigned bits = 0;
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const unsigned long long h = slam_mix64(seed ^ (((base >> 2) + q) * 0xD1342543DE82EF95ull));
#pragma unroll
    for (int e = 0; e < 4; e++) bits |= ((unsigned)((h >> (16 * e)) & 0xFFFFull) >= thresh16 ? 1u : 0u) << (q * 4 + e);
  }
  return bits;
}
static inline unsigned slam_drop_thresh16(float p) {
  const double t = (double)p * 65536.0 + 0.5;
  return t >= 65535.0 ? 65535u : (unsigned)t;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
