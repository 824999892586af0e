// Where the time of the one-launch conv2 -> conv3 of layer1 (cotr_amd/csrc/conv23.hip) goes: the product kernel and its ablations, timed by
// HIP events at B pairs (2048 tiles at 32).
//   ABL 0: the product     1: no identity reads / y writes     2: phase 1 only     4: no barriers / waits in phase 2     8: phase 2 only
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../cotr_amd/csrc conv23_probe.hip -o conv23_probe.exe
//   ./conv23_probe.exe [pairs ...]
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../cotr_amd/csrc/conv23.hip"

thread_local int cotr_tls_device = -1;
static KnobSet g_knobs = {};
thread_local const KnobSet* cotr_tls_knobs = &g_knobs;
static float* g_zero = nullptr;
const float* gemm_zero_buffer() { return g_zero; }

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                           \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

template <int ABL>
static void run(const Conv23Params& p, const char* tag) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv23_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, C23_SMEM));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(conv23_kernel<ABL>, dim3(p.tiles), dim3(256), C23_SMEM, 0, p);
  CK(hipDeviceSynchronize());
  const int iters = 20;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(conv23_kernel<ABL>, dim3(p.tiles), dim3(256), C23_SMEM, 0, p);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters, fl = 2.0 * p.tiles * 128 * 64 * (576 + 256);
  printf("  %-44s %7.1f us/launch  %6.1f TFLOP/s-equivalent (%.3f of 157.3)\n", tag, us, fl / us * 1e-6, fl / us * 1e-6 / 157.3);
}

int main(int argc, char** argv) {
  std::vector<int> pairs;
  for (int i = 1; i < argc; ++i) pairs.push_back(atoi(argv[i]));
  if (pairs.empty()) pairs = {32};
  CK(hipMalloc(&g_zero, 256));
  CK(hipMemset(g_zero, 0, 256));
  for (int B : pairs) {
    const size_t px = (size_t)B * 64 * 128;
    float *t1, *w2, *w3, *par, *res, *y;
    CK(hipMalloc(&t1, px * 64 * 4));
    CK(hipMalloc(&res, px * 256 * 4));
    CK(hipMalloc(&y, px * 256 * 4));
    CK(hipMalloc(&w2, 64 * 576 * 4));
    CK(hipMalloc(&w3, 256 * 64 * 4));
    CK(hipMalloc(&par, 640 * 4));
    std::vector<float> h(px * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 1023) / 1024.f;
    CK(hipMemcpy(t1, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w2, h.data(), 64 * 576 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w3, h.data(), 256 * 64 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(par, h.data(), 640 * 4, hipMemcpyHostToDevice));
    CK(hipMemset(res, 0, px * 256 * 4));
    Conv23Params p;
    p.t1 = t1; p.w2 = w2; p.s2 = par; p.b2 = par + 64; p.w3 = w3; p.s3 = par + 128; p.b3 = par + 384; p.residual = res; p.y = y;
    p.zeros = g_zero;
    p.tiles = B * 64;
    p.stagger = 0;
    printf("%d pairs (%d tiles of 128 pixels, 768 resident):\n", B, p.tiles);
    run<0>(p, "product");
    run<1>(p, "no identity reads / y writes");
    run<2>(p, "phase 1 only (conv2)");
    run<4>(p, "no barriers / waits in phase 2");
    run<5>(p, "no identity / y, no barriers / waits");
    run<8>(p, "phase 2 only (conv3 + identity + y)");
    run<9>(p, "phase 2 only, no identity / y");
    for (int st : {2, 4, 6, 9, 12}) {
      p.stagger = st;
      char tag[64];
      snprintf(tag, sizeof tag, "product, stagger %d (%.0f us per slot)", st, st * 8128 / 2250.0);
      run<0>(p, tag);
    }
    for (int st : {1, 2, 3, 5}) {
      p.stagger = st;
      char tag[64];
      snprintf(tag, sizeof tag, "phase 2 only, stagger %d", st);
      run<8>(p, tag);
    }
    p.stagger = 0;
    CK(hipFree(t1)); CK(hipFree(res)); CK(hipFree(y)); CK(hipFree(w2)); CK(hipFree(w3)); CK(hipFree(par));
  }
  return 0;
}
