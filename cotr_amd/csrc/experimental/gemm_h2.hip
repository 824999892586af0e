// Packing kernel of the split-f16 research path (gemm_h2.h): fp32 -> (f16 hi | f16 lo * 2^11 << 16), one dword per element, and back.
#include "../common.h"
#include "gemm_h2.h"

namespace {
__global__ __launch_bounds__(256) void split_h2_kernel(const float* x, const float* __restrict__ x2, unsigned int* y, const size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    if (x2 != nullptr) v += reinterpret_cast<const f32x4*>(x2)[i];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = h2_pack(v[e]);
    reinterpret_cast<u32x4*>(y)[i] = o;
  }
}
__global__ __launch_bounds__(256) void unsplit_h2_kernel(const unsigned int* __restrict__ x, float* __restrict__ y, const size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const u32x4 v = reinterpret_cast<const u32x4*>(x)[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = h2_unpack(v[e]);
    reinterpret_cast<f32x4*>(y)[i] = o;
  }
}
}  // namespace

// the inverse (exact: hi + lo * 2^-11 is representable in fp32): debug taps and cotr_backbone's feature output
int launch_unsplit_h2(const void* x, float* y, size_t n, hipStream_t s) {
  if (n == 0) return 0;
  if (n % 4 != 0 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return -1;
  const size_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(unsplit_h2_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const unsigned int*>(x), y, n4);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// n floats (a multiple of 4, 16-byte aligned pointers; x == y allowed)
int launch_split_h2(const float* x, void* y, size_t n, hipStream_t s, const float* x2) {
  if (n == 0) return 0;
  if (n % 4 != 0 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)x2 & 15)) return -1;
  const size_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(split_h2_kernel, dim3(blocks), dim3(256), 0, s, x, x2, static_cast<unsigned int*>(y), n4);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
