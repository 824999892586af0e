// Micro-benchmark: what fp32-MFMA rate and what shader clock does the MI355X sustain with NOTHING but MFMAs in flight - on zeros and
// on random operands?  (The chip clocks to its power budget: the roofline's 157.3 TFLOP/s assumes 2.4 GHz.)  Every wavefront runs
// `iters` rounds of 4 independent v_mfma_f32_32x32x2_f32 chains x 16; wave 0 of every workgroup stamps s_memtime (shader cycles) and
// the 100 MHz wall clock before and after.   hipcc --offload-arch=gfx950 -O3 mfma_clock.hip -o mfma_clock.exe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_mfma(const float* in, float* out, unsigned long long* stamps, int iters) {
  const int t = threadIdx.x;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = in[(t * 8 + i) & 4095]; b[i] = in[(t * 8 + i + 2048) & 4095]; }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + i) & 7], b[(k * 2 + i) & 7], acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + t] = s;
  if (t == 0) { stamps[blockIdx.x * 2] = c1 - c0; stamps[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main() {
  const int wgs = 256 * 2, iters = 20000;
  float *in, *out; unsigned long long* st;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, wgs * 256 * 4); hipMalloc(&st, wgs * 16);
  std::vector<float> h(4096);
  for (int mode = 0; mode < 3; ++mode) {
    for (auto& v : h) v = mode == 0 ? 0.f : (mode == 1 ? (float)rand() / RAND_MAX - 0.5f : 1.0f);
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_mfma, dim3(wgs), dim3(256), 0, 0, in, out, st, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> s(wgs * 2);
      hipMemcpy(s.data(), st, wgs * 16, hipMemcpyDeviceToHost);
      double cyc = 0, wall = 0;
      for (int i = 0; i < wgs; ++i) { cyc += s[i * 2]; wall += s[i * 2 + 1]; }
      const double flop = (double)wgs * 4 /*waves*/ * iters * 16 * (2.0 * 32 * 32 * 2);
      printf("%-7s rep %d: %7.2f ms  %6.1f TFLOP/s  shader clock %.0f MHz  (cycles per MFMA per wave %.1f)\n",
             mode == 0 ? "zeros" : mode == 1 ? "random" : "ones", rep, ms, flop / ms / 1e9, cyc / wall * 100.0, cyc / wgs / (iters * 16.0));
    }
  }
  return 0;
}
