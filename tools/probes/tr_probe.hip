// tools only (not part of libslamhip.so): what ds_read_b64_tr_b16 returns for arbitrary per-lane addresses, and what LDS layouts of a
// row-major bf16 tile cost when its transposed MFMA operand is fetched with it.  Built by tools/probes/build.sh into tools/probes/tr_probe.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }

// ---- semantics: lds[i] = i (u16), lane l reads at byte offset addr[l]; out[l*4 + j] = element j of lane l's result
extern "C" __global__ void tr_sem_kernel(const unsigned* addr, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned a = lds_off(lds) + addr[threadIdx.x];
  u32x2_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = (unsigned short)(r[0] & 0xffff);
  out[threadIdx.x * 4 + 1] = (unsigned short)(r[0] >> 16);
  out[threadIdx.x * 4 + 2] = (unsigned short)(r[1] & 0xffff);
  out[threadIdx.x * 4 + 3] = (unsigned short)(r[1] >> 16);
}

// ---- timing: every wave issues `iters` x 16 reads of the form the attention kernels would use, addresses from a table
// kind 0: ds_read_b64_tr_b16, kind 1: ds_read_b128, kind 2: ds_read_b64
template <int KIND>
__global__ void tr_time_kernel(const unsigned* addr /*[16][64] byte offsets*/, int iters, unsigned long long* cycles, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(dyn)[i] = i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned a[16];
#pragma unroll
  for (int k = 0; k < 16; k++) a[k] = lds_off(dyn) + addr[k * 64 + lane];
  unsigned acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if constexpr (KIND == 1) {
      u32x4_t r[16];
#pragma unroll
      for (int k = 0; k < 16; k++) asm volatile("ds_read_b128 %0, %1" : "=v"(r[k]) : "v"(a[k]));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 16; k++) { asm volatile("" : "+v"(r[k])); acc ^= r[k][0] ^ r[k][3]; }
    } else {
      u32x2_t r[16];
#pragma unroll
      for (int k = 0; k < 16; k++) {
        if constexpr (KIND == 0) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r[k]) : "v"(a[k]));
        else asm volatile("ds_read_b64 %0, %1" : "=v"(r[k]) : "v"(a[k]));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 16; k++) { asm volatile("" : "+v"(r[k])); acc ^= r[k][0] ^ r[k][1]; }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

extern "C" int tr_sem(const unsigned* addr, unsigned short* out, void* stream) {
  hipLaunchKernelGGL(tr_sem_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, addr, out);
  return (int)hipGetLastError();
}
extern "C" int tr_time(int kind, const unsigned* addr, int iters, int waves, int blocks, unsigned long long* cycles, unsigned* sink, void* stream) {
  auto k = kind == 0 ? tr_time_kernel<0> : (kind == 1 ? tr_time_kernel<1> : tr_time_kernel<2>);
  static bool set[3] = {false, false, false};
  if (!set[kind]) { hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536); set[kind] = true; }
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 65536, (hipStream_t)stream, addr, iters, cycles, sink);
  return (int)hipGetLastError();
}
