// Shader-clock probe: ONE wavefront that runs beside un-instrumented production kernels (on its own stream) and samples the shader
// clock - s_memtime (shader cycles) against s_memrealtime (100 MHz wall clock) over windows of `window` wall ticks - so that the
// clock the GEMM / attention loops really run at can be read without touching them.  Built as a tiny shared library
// (tools/clock_settle.py loads it with ctypes):
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC clock_probe.hip -o libclock_probe.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, int samples, int window, const int* stop) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < samples; ++i) {
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    unsigned long long w1;
    do {
      __builtin_amdgcn_s_sleep(32);
      w1 = wall_clock64();
    } while (w1 - w0 < (unsigned long long)window);
    const unsigned long long c1 = __builtin_readcyclecounter();
    out[i * 3] = c1 - c0;
    out[i * 3 + 1] = w1 - w0;
    out[i * 3 + 2] = w0;
    if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
      for (int j = i + 1; j < samples; ++j) out[j * 3] = out[j * 3 + 1] = out[j * 3 + 2] = 0;
      return;
    }
  }
}

extern "C" int clock_probe_launch(unsigned long long* out, int samples, int window, const int* stop, void* stream) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out, samples, window, stop);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// device wall-clock now (to align host events with probe samples is not needed: the windows carry their own start ticks)

// ---- L2 warmer probe (tools/l2_warm_probe.py): every XCD reads the whole region once (workgroup b runs on XCD b % 8 and touches
// slice b / 8 of the region, one 128-B line per lane and load), result discarded ----
__global__ __launch_bounds__(256) void l2_touch_kernel(const float* __restrict__ p, size_t bytes, float* sink) {
  const int xcd_slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const size_t lines = bytes / 128, per = (lines + slots - 1) / slots;
  const size_t l0 = (size_t)xcd_slot * per, l1 = l0 + per < lines ? l0 + per : lines;
  float acc = 0.f;
  for (size_t l = l0 + threadIdx.x; l < l1; l += 256) acc += p[l * 32];
  if (acc == 123.456f) sink[0] = acc;   // never true: keeps the loads
}
extern "C" int l2_touch_launch(const float* p, size_t bytes, int wgs_per_xcd, float* sink, void* stream) {
  hipLaunchKernelGGL(l2_touch_kernel, dim3(8 * wgs_per_xcd), dim3(256), 0, static_cast<hipStream_t>(stream), p, bytes, sink);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
