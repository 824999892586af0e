// Where the time of the one-launch FFN block (cotr_amd/csrc/ffn_rows.hip) goes: the product kernel and its ablations, timed by HIP events
// and by phase stamps of wavefront 0 (shader cycles + the 100 MHz wall clock -> the clock the kernel really runs at).
//   ABL 4: the product stream + stamps     5: no refill DMA (stale weights: timing only)     6: no vmcnt waits (timing only)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../cotr_amd/csrc ffn_rows_probe.hip -o ffn_rows_probe.exe
//   ./ffn_rows_probe.exe [tiles ...]          (a tile = 64 rows; 256 tiles = one per CU)
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../cotr_amd/csrc/ffn_rows.hip"

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
static void run(const FfnRowsParams& p0, int tiles, unsigned long long* dbg_d, const char* tag) {
  FfnRowsParams p = p0;
  p.M = tiles * FR_BM;
  p.dbg = dbg_d;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_rows_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFfnRowsSmem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(ffn_rows_kernel<ABL>, dim3(tiles), dim3(256), kFfnRowsSmem, 0, p);
  CK(hipDeviceSynchronize());
  const int iters = 10;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(ffn_rows_kernel<ABL>, dim3(tiles), dim3(256), kFfnRowsSmem, 0, p);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> st((size_t)tiles * 24);
  CK(hipMemcpy(st.data(), dbg_d, st.size() * 8, hipMemcpyDeviceToHost));
  double cyc[11] = {}, wall[11] = {};
  for (int w = 0; w < tiles; ++w)
    for (int k = 0; k < 11; ++k) {
      cyc[k] += (double)(st[((size_t)w * 12 + k + 1) * 2] - st[((size_t)w * 12 + k) * 2]) / tiles;
      wall[k] += (double)(st[((size_t)w * 12 + k + 1) * 2 + 1] - st[((size_t)w * 12 + k) * 2 + 1]) / tiles;
    }
  double tc = 0, tw = 0;
  for (int k = 0; k < 11; ++k) { tc += cyc[k]; tw += wall[k]; }
  const double us = ms * 1e3 / iters, fl = 2.0 * 2 * 256 * 1024 * p.M;
  printf("%-22s tiles %4d: %7.1f us/launch  %6.1f TFLOP/s (%.3f of 157.3) | wave 0 of a tile: %.0f cycles in %.2f us = %.3f GHz\n", tag, tiles, us,
         fl / us * 1e-6, fl / us * 1e-6 / 157.3, tc, tw / 100.0, tc / (tw * 10.0));
  printf("    cycles: prologue %.0f | blocks", cyc[0]);
  for (int k = 1; k <= 8; ++k) printf(" %.0f", cyc[k]);
  printf(" (ideal 32768 each) | wait for the others %.0f | epilogue %.0f\n", cyc[9], cyc[10]);
}

int main(int argc, char** argv) {
  std::vector<int> tiles;
  for (int i = 1; i < argc; ++i) tiles.push_back(atoi(argv[i]));
  if (tiles.empty()) tiles = {64, 256, 512};
  const int maxM = 1024 * FR_BM;
  float *X, *W1, *b1, *W2, *b2, *lw, *lb, *Y;
  unsigned long long* dbg;
  CK(hipMalloc(&X, (size_t)maxM * 256 * 4));
  CK(hipMalloc(&Y, (size_t)maxM * 256 * 4));
  CK(hipMalloc(&W1, 1024 * 256 * 4));
  CK(hipMalloc(&W2, 1024 * 256 * 4));
  CK(hipMalloc(&b1, 1024 * 4));
  CK(hipMalloc(&b2, 256 * 4));
  CK(hipMalloc(&lw, 256 * 4));
  CK(hipMalloc(&lb, 256 * 4));
  CK(hipMalloc(&g_zero, 256));
  CK(hipMemset(g_zero, 0, 256));
  CK(hipMalloc(&dbg, (size_t)1024 * 24 * 8));
  std::vector<float> h((size_t)maxM * 256);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) * (1.f / 65536.f) - 0.5f; };
  for (auto& v : h) v = rnd();
  CK(hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (size_t i = 0; i < 1024 * 256; ++i) h[i] = rnd() * 0.1f;
  CK(hipMemcpy(W1, h.data(), 1024 * 256 * 4, hipMemcpyHostToDevice));
  for (size_t i = 0; i < 1024 * 256; ++i) h[i] = rnd() * 0.1f;
  CK(hipMemcpy(W2, h.data(), 1024 * 256 * 4, hipMemcpyHostToDevice));
  for (size_t i = 0; i < 1024; ++i) h[i] = rnd() * 0.1f;
  CK(hipMemcpy(b1, h.data(), 1024 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b2, h.data(), 256 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(lb, h.data(), 256 * 4, hipMemcpyHostToDevice));
  for (size_t i = 0; i < 256; ++i) h[i] = 1.f + rnd() * 0.1f;
  CK(hipMemcpy(lw, h.data(), 256 * 4, hipMemcpyHostToDevice));
  FfnRowsParams p = {};
  p.X = X; p.W1 = W1; p.b1 = b1; p.W2 = W2; p.b2 = b2; p.ln_w = lw; p.ln_b = lb; p.Y = Y; p.zeros = g_zero;
  for (int t : tiles) {
    if (t > 1024) continue;
    run<4>(p, t, dbg, "product + stamps");
    run<5>(p, t, dbg, "no refill DMA");
    run<6>(p, t, dbg, "no vmcnt waits");
  }
  return 0;
}
