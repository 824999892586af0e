// Ablation micro-benchmark of the k-split GEMM (COTR_ABL: 0 full, 1 no global loads after the first, 2 no MFMA)
#include "../../cotr_amd/csrc/gemm.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <stdlib.h>
template <typename F> float run(F launch, int n) {
  hipStream_t s; (void)hipStreamCreate(&s);
  hipGraph_t g; hipGraphExec_t e;
  for (int i = 0; i < 3; ++i) launch(s, i);
  (void)hipStreamSynchronize(s);
  (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) launch(s, i);
  (void)hipStreamEndCapture(s, &g); (void)hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipGraphLaunch(e, s); (void)hipStreamSynchronize(s);
  float best = 1e9;
  for (int r = 0; r < 5; ++r) { (void)hipEventRecord(a, s); (void)hipGraphLaunch(e, s); (void)hipEventRecord(b, s); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
  return best * 1000.f / n;
}
int main() {
  const int Mmax = 1024, N = 256, Kmax = 2304;
  float *x, *w, *y; (void)hipMalloc(&x, (size_t)Mmax * Kmax * 4); (void)hipMalloc(&w, (size_t)N * Kmax * 4); (void)hipMalloc(&y, (size_t)Mmax * N * 4);
  std::vector<float> h((size_t)Mmax * Kmax); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(w, h.data(), (size_t)N * Kmax * 4, hipMemcpyHostToDevice);
  const char* big = getenv("ABL_BIG");
  if (big) {
    const int M = 16384;
    float *xb, *yb; (void)hipMalloc(&xb, (size_t)M * Kmax * 4); (void)hipMalloc(&yb, (size_t)M * N * 4);
    for (size_t off = 0; off < (size_t)M * Kmax; off += h.size()) (void)hipMemcpy(xb + off, h.data(), (std::min(h.size(), (size_t)M * Kmax - off)) * 4, hipMemcpyHostToDevice);
    for (int K : {256, 1024, 2304}) for (int cfg : {0, 1, 2}) {
      GemmParams p; memset(&p, 0, sizeof(p)); p.colscale = 1.f; p.a2_period = 1;
      p.M = M; p.N = N; p.K = K; p.A = xb; p.lda = K; p.W = w; p.C = yb; p.ldc = N;
      float us = run([&](hipStream_t s, int) { launch_gemm_cfg(GEMM_DENSE, cfg, p, s); }, 20);
      printf("ABL=%d M=%5d K=%4d cfg=%2d: %.2f us  (%.1f TFLOP/s)\n", COTR_ABL, M, K, cfg, us, 2.0 * M * N * K / us * 1e-6);
    }
    return 0;
  }
  const char* occ = getenv("ABL_OCC");
  if (occ) {  // does a second co-resident workgroup per CU overlap ingest with MFMA?  (256 vs 512 vs 1024 workgroups)
    float *xb, *yb; (void)hipMalloc(&xb, (size_t)4096 * Kmax * 4); (void)hipMalloc(&yb, (size_t)4096 * N * 4);
    for (size_t off = 0; off < (size_t)4096 * Kmax; off += h.size()) (void)hipMemcpy(xb + off, h.data(), (std::min(h.size(), (size_t)4096 * Kmax - off)) * 4, hipMemcpyHostToDevice);
    for (int K : {1024, 2304}) for (int cfg : {3, 4}) for (int M : {512, 1024, 2048, 4096}) {
      GemmParams p; memset(&p, 0, sizeof(p)); p.colscale = 1.f; p.a2_period = 1;
      p.M = M; p.N = N; p.K = K; p.A = xb; p.lda = K; p.W = w; p.C = yb; p.ldc = N;
      float us = run([&](hipStream_t s, int) { launch_gemm_cfg(GEMM_DENSE, cfg, p, s); }, 30);
      printf("OCC cfg=%d K=%4d M=%4d (%4d WGs): %.2f us\n", cfg, K, M, (M / 32) * (N / 32), us);
    }
    return 0;
  }
  for (int M : {32, 512}) for (int K : {256, 1024, 2304}) for (int cfg : {3, 13, 4, 14, 8}) {
    GemmParams p; memset(&p, 0, sizeof(p)); p.colscale = 1.f; p.a2_period = 1;
    p.M = M; p.N = N; p.K = K; p.A = x; p.lda = K; p.W = w; p.C = y; p.ldc = N;
    float us = run([&](hipStream_t s, int) { launch_gemm_cfg(GEMM_DENSE, cfg, p, s); }, 50);
    printf("ABL=%d M=%4d K=%4d cfg=%2d: %.2f us\n", COTR_ABL, M, K, cfg, us);
  }
  return 0;
}
