// Micro-benchmark: how long does it take the dispatcher to get all workgroups of a small grid running, as a function of the
// workgroup shape (waves), its LDS allocation and its register allocation?  Every workgroup stamps the 100 MHz wall clock at its
// first instruction; printed: mean / max offset to the first workgroup's stamp, and the launch-to-launch time of a dependent
// chain of such kernels (graph-captured).  gfx950.   hipcc --offload-arch=gfx950 -O3 launch_ramp.hip -o launch_ramp.exe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

template <int NT, int VG>
__global__ __launch_bounds__(NT) void k_stamp(unsigned long long* out, float* sink, int spin) {
  extern __shared__ float lds[];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t0;
  // keep VG registers alive (forces the allocation) and optionally do a little work
  float v[VG];
#pragma unroll
  for (int i = 0; i < VG; ++i) v[i] = (float)(threadIdx.x + i);
  for (int s = 0; s < spin; ++s) {
#pragma unroll
    for (int i = 0; i < VG; ++i) v[i] = v[i] * 1.0001f + 0.5f;
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < VG; ++i) acc += v[i];
  if (acc == 12345.678f) { lds[threadIdx.x] = acc; sink[threadIdx.x] = lds[(threadIdx.x + 1) % NT]; }
}

template <int NT, int VG>
void run(const char* name, int wgs, size_t lds, unsigned long long* d_out, float* sink, bool detail = false) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_stamp<NT, VG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipStream_t s; hipStreamCreate(&s);
  std::vector<unsigned long long> h(wgs);
  double mean = 0, mx = 0;
  for (int rep = 0; rep < 6; ++rep) {
    hipMemsetAsync(d_out, 0, wgs * 8, s);
    hipStreamSynchronize(s);
    hipLaunchKernelGGL((k_stamp<NT, VG>), dim3(wgs), dim3(NT), lds, s, d_out, sink, 0);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d_out, wgs * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = *std::min_element(h.begin(), h.end());
    mean = 0; mx = 0;
    for (auto t : h) { double d = (double)(t - t0) * 0.01; mean += d; mx = std::max(mx, d); }
    mean /= wgs;
    if (rep == 5 && detail) {   // per XCD (workgroup b runs on XCD b % 8): mean entry offset, and by position inside the XCD's share
      double x[8] = {0}; int n[8] = {0};
      for (int b = 0; b < wgs; ++b) { x[b & 7] += (double)(h[b] - t0) * 0.01; n[b & 7]++; }
      printf("    per XCD mean entry offset:");
      for (int i = 0; i < 8; ++i) printf(" %5.2f", x[i] / n[i]);
      printf("\n    by dispatch position (b / 8) for XCD 0:");
      for (int b = 0; b < wgs; b += 8) printf(" %4.2f", (double)(h[b] - t0) * 0.01);
      printf("\n");
    }
  }
  // dependent chain, graph captured
  hipGraph_t g; hipGraphExec_t e;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((k_stamp<NT, VG>), dim3(wgs), dim3(NT), lds, s, d_out, sink, 0);
  hipStreamEndCapture(s, &g); hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipGraphLaunch(e, s); hipStreamSynchronize(s);
  float best = 1e9;
  for (int r = 0; r < 5; ++r) { hipEventRecord(a, s); hipGraphLaunch(e, s); hipEventRecord(b, s); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); best = std::min(best, ms); }
  printf("%-28s wgs %4d  threads %4d  lds %6zu  entry offset mean %5.2f max %5.2f us | chain %5.2f us/launch\n", name, wgs, NT, lds, mean, mx,
         best * 10.f);
  hipGraphExecDestroy(e); hipGraphDestroy(g); hipStreamDestroy(s);
}

int main() {
  unsigned long long* d_out; float* sink;
  hipMalloc(&d_out, 8 * 8192); hipMalloc(&sink, 4096 * 4);
  for (size_t lds : {(size_t)0, (size_t)32768, (size_t)65536, (size_t)133120, (size_t)163840}) {
    run<512, 8>("8 waves, few regs", 256, lds, d_out, sink);
    run<256, 8>("4 waves, few regs", 256, lds, d_out, sink);
    run<64, 8>("1 wave, few regs", 256, lds, d_out, sink);
  }
  run<512, 8>("8 waves, few regs", 256, 65536, d_out, sink, true);
  run<64, 8>("1 wave, few regs", 64, 0, d_out, sink, true);
  run<64, 8>("1 wave, few regs", 8, 0, d_out, sink, true);
  run<512, 64>("8 waves, 64+ regs", 256, 65536, d_out, sink);
  run<512, 120>("8 waves, 120+ regs", 256, 65536, d_out, sink);
  run<256, 120>("4 waves, 120+ regs", 256, 65536, d_out, sink);
  run<256, 8>("4 waves, few regs", 512, 65536, d_out, sink);
  run<256, 8>("4 waves, few regs", 1024, 32768, d_out, sink);
  run<512, 8>("8 waves, few regs", 512, 65536, d_out, sink);
  run<512, 8>("8 waves, few regs", 128, 65536, d_out, sink);
  run<1024, 8>("16 waves, few regs", 256, 65536, d_out, sink);
  run<1024, 8>("16 waves, few regs", 128, 65536, d_out, sink);
  return 0;
}
