// Packing kernel of the split-f16 research path (gemm_h2.h): fp32 -> (f16 hi | f16 lo * 2^11 << 16), one dword per element, and back.
#include "../common.h"
#include "gemm_h2.h"

// one int per device: raised by any kernel that packs an activation outside f16's range (gemm_h2.h: h2_pack_chk)
int* h2_overflow_flag() {
  static int* flags[COTR_MAX_DEVICES] = {};
  int*& f = flags[cotr_current_device()];
  if (f == nullptr) {
    int* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), 64) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 64) != hipSuccess) { (void)hipFree(p); return nullptr; }
    f = p;
  }
  return f;
}

namespace {
__global__ __launch_bounds__(256) void split_h2_kernel(const float* x, const float* __restrict__ x2, unsigned int* y, const size_t n4, int* ovf) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    if (x2 != nullptr) v += reinterpret_cast<const f32x4*>(x2)[i];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = h2_pack_chk(v[e], ovf);
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
// layernorm_kernel (pointwise.hip: the same arithmetic in the same order, so y has the same bits) that ALSO leaves the packed split-f16
// form of y (+ add, the decoder's query_pos) for the projection that consumes it - instead of a packing launch in between
__device__ __forceinline__ float h2_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__global__ __launch_bounds__(256) void layernorm_h2_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                           float* __restrict__ y, unsigned int* __restrict__ yp, const float* __restrict__ add,
                                                           int rows, int* ovf) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * 256 + lane * 4);
  const float mean = h2_wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
  const f32x4 d = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
  const float var = h2_wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256.f);
  const float rstd = 1.f / sqrtf(var + 1e-5f);
  const f32x4 ww = *reinterpret_cast<const f32x4*>(w + lane * 4);
  const f32x4 bb = *reinterpret_cast<const f32x4*>(b + lane * 4);
  f32x4 out;
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = d[i] * rstd * ww[i] + bb[i];
  *reinterpret_cast<f32x4*>(y + (size_t)row * 256 + lane * 4) = out;
  if (add != nullptr) out += *reinterpret_cast<const f32x4*>(add + (size_t)row * 256 + lane * 4);
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = h2_pack_chk(out[i], ovf);
  *reinterpret_cast<u32x4*>(yp + (size_t)row * 256 + lane * 4) = o;
}
}  // namespace

int launch_layernorm_h2(const float* x, const float* w, const float* b, float* y, void* yp, const float* add, int rows, hipStream_t s) {
  if (rows <= 0) return 0;
  int* ovf = h2_overflow_flag();
  if (ovf == nullptr) return -2;
  hipLaunchKernelGGL(layernorm_h2_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, y, static_cast<unsigned int*>(yp), add, rows, ovf);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

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
  int* ovf = h2_overflow_flag();
  if (ovf == nullptr) return -2;
  hipLaunchKernelGGL(split_h2_kernel, dim3(blocks), dim3(256), 0, s, x, x2, static_cast<unsigned int*>(y), n4, ovf);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
