// Where the time of expand.hip goes (layer1 block 0's downsample + conv1 at 32 pairs: 262144 rows, K = 64 -> 256 + 64 channels):
// the product kernel and its ablations by HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../cotr_amd/csrc expand_probe.hip -o expand_probe.exe
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../cotr_amd/csrc/expand.hip"

thread_local int cotr_tls_device = -1;
static KnobSet g_knobs = {};
thread_local const KnobSet* cotr_tls_knobs = &g_knobs;
const float* gemm_zero_buffer() { return nullptr; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int ABL>
static void run(const ExpandParams& p, const char* tag) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((expand64_kernel<ABL>), dim3(p.tiles), dim3(256), EX_SMEM, 0, p);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((expand64_kernel<ABL>), dim3(p.tiles), dim3(256), EX_SMEM, 0, p);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("  %-40s stagger %d: %7.1f us/launch\n", tag, p.stagger, ms * 1e3 / 20);
}
int main() {
  const int M = 262144;
  float *x, *w, *par, *y0, *y1;
  CK(hipMalloc(&x, (size_t)M * 64 * 4)); CK(hipMalloc(&w, 320 * 64 * 4)); CK(hipMalloc(&par, 640 * 4));
  CK(hipMalloc(&y0, (size_t)M * 256 * 4)); CK(hipMalloc(&y1, (size_t)M * 64 * 4));
  std::vector<float> h((size_t)M * 64);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 1023) / 1024.f;
  CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data(), 320 * 64 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(par, h.data(), 640 * 4, hipMemcpyHostToDevice));
  ExpandParams p;
  p.x = x;
  p.seg[0] = {w, par, par + 256, y0, 256, 0, 8};
  p.seg[1] = {w + 256 * 64, par + 512, par + 576, y1, 64, 1, 2};
  p.tiles = M / 128;
  for (int st : {0, 2, 5}) {
    p.stagger = st;
    run<0>(p, "product");
    run<1>(p, "no stores");
    run<2>(p, "no matrix instructions");
    run<3>(p, "neither");
    run<4>(p, "no barriers");
  }
  return 0;
}
